// 'fat' edge gate on the bf16 matrix cores for the head geometries other than the shipped 8 x (64, 64, 32): the bf16 twin of
// edge_gate_heads.hip, with the algebra, operand construction and lane model of edge_gate_bf16.hip (reference
// network_MMG.py:96-104; NUM_HEADS / DIM_ATTEN: network_MMG.py:48-50):
//   per (edge, head) row: hidden = relu(Gq[src] + W0k . kproj_row), logits = W3 . hidden + b3, prob = softmax over the d_o
//   channels, gated = prob * value[dst]; a wave owns 32 consecutive edges of one head; both layers as transposed
//   v_mfma_f32_32x32x16_bf16 products (TERMS = 3: operands as bf16 hi + lo, three MFMAs per product; 1: single rounding);
//   the hidden layer goes from the layer-1 accumulator registers straight into the layer-2 product (k-slot (half hi, element
//   e) of step (to, half) is o = 32 to + 16 half + 8 (e >> 2) + 4 hi + (e & 3): what registers 8 half + e hold).
// Template: d_k in {32, 64, 128} (d_k / 16 k-steps in layer 1, 2 d_k / 32 hidden slices), d_o = DIM_ATTEN / heads (ceil(d_o / 32)
// logit blocks, rows of W3 past d_o zero, their channels masked out of the softmax).  bf16 planes of W0k [2 d_k][d_k] and
// W3 [d_o][2 d_k] in LDS (row pitches 2 d_k + 16 and 4 d_k + 8 bytes as in edge_gate_bf16.hip); at d_k = 128 they take
// 103 KB per plane set: one 8-wave block per CU, and no split-bf16 variant (two plane sets do not fit) -- that combination
// stays on the fp32 kernel (launch returns 1).
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KS: format of kproj -- 0 fp32, 1 split-pair words, 2 half rows (bf16; TERMS = 1)
template <int TERMS, int KS, int DK, int DOX>
__global__ __launch_bounds__(DK == 128 ? 512 : 256, DK == 128 ? 1 : 2) void edge_gate_bf16_hd_kernel(GateArgs p, int n_heads) {
    constexpr int PL = TERMS == 1 ? 1 : 2;
    constexpr bool F16 = KS == 3;                 // KS 3: fp16 half rows and fp16 operands (precision mode fp16_mixed; TERMS = 1), as in edge_gate_bf16.hip
    auto cv4 = [](const f32x4& x) {
        if constexpr (F16) {
            typedef _Float16 f16x4_g __attribute__((ext_vector_type(4)));
            f32x4 y;
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = __builtin_amdgcn_fmed3f(x[c], -65504.f, 65504.f);
            return __builtin_bit_cast(bf16x4, __builtin_convertvector(y, f16x4_g));
        } else {
            return __builtin_convertvector(x, bf16x4);
        }
    };
    constexpr int HID = 2 * DK, TO = HID / 32, MO = (DOX + 31) / 32, NK1 = DK / 16;
    constexpr int P0 = 2 * DK + 16, P3 = 2 * HID + 8;        // plane row pitches in bytes
    constexpr int W0B = HID * P0, W3B = MO * 32 * P3;
    constexpr int NT = DK == 128 ? 512 : 256, NW = NT / 64;
    static_assert(DOX % 8 == 0 && PL * (W0B + W3B) <= 160 * 1024, "gate geometry");
    __shared__ __attribute__((aligned(16))) char smem[PL * (W0B + W3B)];
    char* sW0 = smem;                    // [PL][HID][P0]
    char* sW3 = smem + PL * W0B;         // [PL][MO * 32][P3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int A = n_heads * DOX;

    for (int i = tid; i < HID * (DK / 4); i += NT) {      // four fp32 -> four bf16 (8 B) per plane
        const int r = i / (DK / 4), c4 = (i % (DK / 4)) * 4;
        const f32x4 x = *reinterpret_cast<const f32x4*>(p.w0k + r * DK + c4);
        const bf16x4 h = cv4(x);
        *reinterpret_cast<bf16x4*>(sW0 + r * P0 + c4 * 2) = h;
        if (PL == 2) *reinterpret_cast<bf16x4*>(sW0 + W0B + r * P0 + c4 * 2) = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), bf16x4);
    }
    for (int i = tid; i < MO * 32 * (HID / 4); i += NT) {
        const int r = i / (HID / 4), c4 = (i % (HID / 4)) * 4;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (r < DOX) x = *reinterpret_cast<const f32x4*>(p.w3 + r * HID + c4);
        const bf16x4 h = cv4(x);
        *reinterpret_cast<bf16x4*>(sW3 + r * P3 + c4 * 2) = h;
        if (PL == 2) *reinterpret_cast<bf16x4*>(sW3 + W3B + r * P3 + c4 * 2) = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), bf16x4);
    }
    __syncthreads();

    const long n_wu = (long)((p.n_edges + 31) / 32) * n_heads;          // wave units: (block of 32 edges, head)
    for (long u = blockIdx.x; u * NW < n_wu; u += gridDim.x) {
        asm volatile("" ::: "memory");                    // keep the weight fragments out of LICM's hands (edge_gate.hip)
        const long wu = u * NW + wave;
        if (wu >= n_wu) continue;                         // (no barrier in this loop)
        const int h = (int)(wu % n_heads);
        const int e_raw = (int)(wu / n_heads) * 32 + li;
        const bool valid = e_raw < p.n_edges;
        const int e = valid ? e_raw : p.n_edges - 1;
        // ---- this row's kproj values: k-slot (hi, e) of step ks is c = 16 ks + 8 hi + e ----
        bf16x8 zh[NK1], zl[NK1];
        if (p.use_edge) {
            const float* zrow = p.kproj + (size_t)e * 512 + h * DK + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < NK1; ++ks) {
                if (KS >= 2) {                            // eight bf16 (fp16) = one 16-byte load
                    zh[ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(p.kproj + (size_t)e * 512) + (h * DK + 8 * hi + 16 * ks) * 2);
                    zl[ks] = zh[ks];
                    continue;
                }
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(zrow + 16 * ks), x1 = *reinterpret_cast<const f32x4*>(zrow + 16 * ks + 4);
                if (KS == 1) {
                    const u32x4 a = __builtin_bit_cast(u32x4, x0), b = __builtin_bit_cast(u32x4, x1);
                    u32x4 hh, ll;
                    hh[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302u); hh[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302u);
                    hh[2] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u); hh[3] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
                    ll[0] = __builtin_amdgcn_perm(a[1], a[0], 0x05040100u); ll[1] = __builtin_amdgcn_perm(a[3], a[2], 0x05040100u);
                    ll[2] = __builtin_amdgcn_perm(b[1], b[0], 0x05040100u); ll[3] = __builtin_amdgcn_perm(b[3], b[2], 0x05040100u);
                    zh[ks] = __builtin_bit_cast(bf16x8, hh);
                    zl[ks] = __builtin_bit_cast(bf16x8, ll);
                } else {
                    const bf16x4 h0 = __builtin_convertvector(x0, bf16x4), h1 = __builtin_convertvector(x1, bf16x4);
                    zh[ks] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x4 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f32x4), bf16x4);
                    const bf16x4 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f32x4), bf16x4);
                    zl[ks] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
        }
        const int sn = p.src[e], dn = p.dst[e];
        const float* gq = p.node + (size_t)sn * p.ld_node + p.gq_off + h * HID + 4 * hi;
        f32x16 lg[MO];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {              // b3[m], m = mo*32 + 8*r4 + 4*hi + c (zero past d_o)
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (mo * 32 + 8 * r4 < DOX) b = *reinterpret_cast<const f32x4*>(p.b3 + mo * 32 + 8 * r4 + 4 * hi);
#pragma unroll
                for (int c = 0; c < 4; ++c) lg[mo][r4 * 4 + c] = b[c];
            }
#pragma unroll(DK == 128 ? 1 : TO)
        for (int to = 0; to < TO; ++to) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (p.use_edge) {                             // (USE_GCN_EDGE=false: hidden = relu(Gq), the edge half is absent)
                const char* ap = sW0 + (to * 32 + li) * P0 + 16 * hi;
#pragma unroll
                for (int ks = 0; ks < NK1; ++ks) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + 32 * ks);
                    if (PL == 2) {
                        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + W0B + 32 * ks);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, zh[ks], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, zl[ks], acc, 0, 0, 0);
                    }
                    acc = mfma_h<F16>(ah, zh[ks], acc);
                }
            }
            // hidden = relu(acc + Gq[src, h*HID + o]),  o = to*32 + 8*r4 + 4*hi + c  (registers r = 4 r4 + c)
            float hid[16];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 gqv = *reinterpret_cast<const f32x4*>(gq + to * 32 + 8 * r4);
#pragma unroll
                for (int c = 0; c < 4; ++c) hid[r4 * 4 + c] = fmaxf(acc[r4 * 4 + c] + gqv[c], 0.f);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 p0, p1;
#pragma unroll
                for (int c = 0; c < 4; ++c) { p0[c] = hid[8 * half + c]; p1[c] = hid[8 * half + 4 + c]; }
                const bf16x4 h0 = cv4(p0), h1 = cv4(p1);
                const bf16x8 hh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf16x8 hl = hh;
                if (PL == 2) {
                    const bf16x4 l0 = __builtin_convertvector(p0 - __builtin_convertvector(h0, f32x4), bf16x4);
                    const bf16x4 l1 = __builtin_convertvector(p1 - __builtin_convertvector(h1, f32x4), bf16x4);
                    hl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int mo = 0; mo < MO; ++mo) {
                    // W3[m = mo*32 + li][o = to*32 + 16 half + 4 hi + {0..3}] and the same + 8
                    const char* wp = sW3 + (mo * 32 + li) * P3 + (to * 32 + 16 * half + 4 * hi) * 2;
                    const bf16x4 wa = *reinterpret_cast<const bf16x4*>(wp), wb = *reinterpret_cast<const bf16x4*>(wp + 16);
                    const bf16x8 wh = __builtin_shufflevector(wa, wb, 0, 1, 2, 3, 4, 5, 6, 7);
                    if (PL == 2) {
                        const bf16x4 la = *reinterpret_cast<const bf16x4*>(wp + W3B), lb = *reinterpret_cast<const bf16x4*>(wp + W3B + 16);
                        const bf16x8 wl = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
                        lg[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, hh, lg[mo], 0, 0, 0);
                        lg[mo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, hl, lg[mo], 0, 0, 0);
                    }
                    lg[mo] = mfma_h<F16>(wh, hh, lg[mo]);
                }
            }
        }
        // softmax over the d_o channels m = mo*32 + crow32(r, hi) (+ the other 16 of a block in lane^32), times value
        float mx = -INFINITY;
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mo * 32 + 8 * (r >> 2) < DOX) mx = fmaxf(mx, lg[mo][r]);
        mx = half_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mo * 32 + 8 * (r >> 2) < DOX) {
                    lg[mo][r] = __expf(lg[mo][r] - mx);
                    sum += lg[mo][r];
                }
        sum = half_sum(sum);
        const float inv = 1.f / sum;
        if (valid) {
            const float* vrow = p.node + (size_t)dn * p.ld_node + p.v_off + h * DOX + 4 * hi;
            float* grow = p.gated + (size_t)e * A + h * DOX + 4 * hi;
#pragma unroll
            for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    if (mo * 32 + 8 * r4 >= DOX) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(vrow + mo * 32 + 8 * r4);
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = lg[mo][r4 * 4 + c] * inv * v[c];
                    *reinterpret_cast<f32x4*>(grow + mo * 32 + 8 * r4) = o;
                }
            if (p.prob) {                      // test tap in the reference's [E, d_o, H] order
#pragma unroll
                for (int mo = 0; mo < MO; ++mo)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (mo * 32 + 8 * (r >> 2) < DOX) p.prob[(size_t)e * A + (mo * 32 + crow32(r, hi)) * n_heads + h] = lg[mo][r] * inv;
            }
        }
    }
}

template <int TERMS, int KS, int DK, int DOX>
int run(const GateArgs& a, int n_heads, hipStream_t s) {
    constexpr int NT = DK == 128 ? 512 : 256, NW = NT / 64;
    const long n_wu = (long)((a.n_edges + 31) / 32) * n_heads, units = (n_wu + NW - 1) / NW;
    const long cap = a.grid_cap > 0 ? a.grid_cap : (DK == 128 ? 256 : 768);      // persistent: the planes are made once per block
    hipLaunchKernelGGL((edge_gate_bf16_hd_kernel<TERMS, KS, DK, DOX>), dim3((unsigned)std::min(units, cap)), dim3(NT), 0, s, a, n_heads);
    return 0;
}

template <int DK, int DOX>
int pick(const GateArgs& a, int n_heads, int terms, int ks, hipStream_t s) {
    if (terms == 3) {
        if constexpr (DK == 128) return 1;                 // (two plane sets do not fit the LDS)
        else return ks ? run<3, 1, DK, DOX>(a, n_heads, s) : run<3, 0, DK, DOX>(a, n_heads, s);
    }
    return ks == 3 ? run<1, 3, DK, DOX>(a, n_heads, s) : ks == 2 ? run<1, 2, DK, DOX>(a, n_heads, s) : ks ? run<1, 1, DK, DOX>(a, n_heads, s) : run<1, 0, DK, DOX>(a, n_heads, s);
}

}  // namespace

bool edge_gate_bf16_heads_supports(int dk, int dox, int terms) {
    const bool geo = (dk == 32 && (dox == 8 || dox == 16 || dox == 32)) || (dk == 64 && (dox == 16 || dox == 32 || dox == 64)) ||
                     (dk == 128 && (dox == 32 || dox == 64 || dox == 128));
    return geo && (terms == 1 || (terms == 3 && dk != 128));
}

// terms = 3 split-bf16 | 1 single-rounded; kproj_split: 0 fp32, 1 split-pair words, 2 half rows (terms = 1 only).
// Returns 1 when the combination is not built (-> edge_gate_heads.hip, the fp32 kernel).
int launch_edge_gate_bf16_heads(const GateArgs& a, int n_heads, int dk, int dox, int terms, int kproj_split, hipStream_t s) {
    if (a.n_edges <= 0) return 0;
    if ((a.ld_node & 3) || (a.gq_off & 3) || (a.v_off & 3)) return fail(-1, "edge_gate: ld_node/gq_off/v_off must be multiples of 4");
    if (terms != 1 && terms != 3) return fail(-1, "edge_gate_bf16: terms must be 1 or 3");
    if (kproj_split >= 2 && terms != 1) return fail(-1, "edge_gate_bf16: half-row kproj needs terms = 1");
    if (n_heads * dk != 512) return 1;
    int r = 1;
#define VLSAT_GH(DK, DOX) if (dk == DK && dox == DOX) r = pick<DK, DOX>(a, n_heads, terms, kproj_split, s)
    VLSAT_GH(32, 8); VLSAT_GH(32, 16); VLSAT_GH(32, 32);
    VLSAT_GH(64, 16); VLSAT_GH(64, 32); VLSAT_GH(64, 64);
    VLSAT_GH(128, 32); VLSAT_GH(128, 64); VLSAT_GH(128, 128);
#undef VLSAT_GH
    if (r) return r;
    VLSAT_LAUNCH_CHECK("edge_gate_bf16_heads");
    return 0;
}

}  // namespace vlsat
