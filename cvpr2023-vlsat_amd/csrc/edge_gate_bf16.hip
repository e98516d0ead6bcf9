// 'fat' edge gate on the bf16 matrix cores (BASELINE configs[2]): the same algebra and data flow as edge_gate.hip --
// reference network_MMG.py:96-104; per (edge, head) row: hidden = relu(Gq[src] + W0k . kproj_row), logits = W3 . hidden
// + b3, prob = softmax over the 32 channels, gated = prob * value[dst] (head-major) -- with both layers on
// v_mfma_f32_32x32x16_bf16.  TERMS = 3: operands as bf16 hi + lo, three MFMAs per product; TERMS = 1: single rounding.
//
// Transposed products, so a lane owns ONE (edge, head) row and the softmax stays in-lane:
//   hidden^T[o][row] : A = W0k planes from LDS (ds_read_b128, pitch 144 B), B = the row's 64 kproj values from HBM,
//                      split to bf16 in registers (two v_perm_b32 per pair when the proj_edge GEMM wrote them in the
//                      split-pair format, KS = true);
//   logits^T[m][row] : B = hidden straight from the layer-1 accumulator registers -- k-slot (half hi, element e) of step
//                      (to, half) is o = 32 to + 16 half + 8 (e>>2) + 4 hi + (e&3), exactly what registers 8 half + e hold,
//                      as in the attention kernel's PV step; A = W3 planes from LDS, two 8-byte reads per operand.
// The bf16 planes of W0k / W3 are made by the block itself from the fp32 weights (they are tiny).
// One wave = 32 rows = 4 edges per step (72 MFMAs in split-bf16, against 192 64-cycle fp32 ones).
#include "gemm_core.h"
#include "gate_agg.h"
#include "kernels.h"

namespace vlsat {

namespace {

constexpr int GB_P0 = 144;      // W0k plane row pitch (64 bf16 + 16 B)
constexpr int GB_P3 = 264;      // W3 plane row pitch (128 bf16 + 8 B: the 32 rows of a ds_read_b64 land on 32 different bank pairs)

// KS: format of kproj -- 0 fp32, 1 split-pair words, 2 half rows (bf16; TERMS = 1)
// TWIN (round 6): two gates on one edge list in one launch, selected by blockIdx.y (edge_gate.hip)
template <int TERMS, int KS, bool TWIN = false>
__global__ __launch_bounds__(256, TERMS == 1 ? 4 : 2) void edge_gate_bf16_kernel(GateArgs pa, GateArgs pb) {
    const GateArgs& p = (TWIN && blockIdx.y != 0) ? pb : pa;
    constexpr int PL = TERMS == 1 ? 1 : 2;
    constexpr bool F16 = KS == 3;                 // KS 3: kproj holds fp16 half rows, the whole gate runs on fp16 operands (precision mode fp16_mixed; TERMS = 1)
    auto cv4 = [](const f32x4& x) {               // four fp32 -> four 16-bit operands (bf16, or fp16 clamped)
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
    constexpr int W0B = 128 * GB_P0, W3B = 32 * GB_P3;
    __shared__ __attribute__((aligned(16))) char smem[PL * (W0B + W3B) + 4 * AG_WAVE_BYTES];     // + the fused aggregation's wave buffers (gate_agg.h)
    char* sW0 = smem;                    // [PL][128][144]
    char* sW3 = smem + PL * W0B;         // [PL][32][272]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;

    for (int i = tid; i < 128 * 16; i += 256) {           // four fp32 -> four bf16 (8 B) per plane
        const int r = i >> 4, c4 = (i & 15) * 4;
        const f32x4 x = *reinterpret_cast<const f32x4*>(p.w0k + r * 64 + c4);
        const bf16x4 h = cv4(x);
        *reinterpret_cast<bf16x4*>(sW0 + r * GB_P0 + c4 * 2) = h;
        if (PL == 2) *reinterpret_cast<bf16x4*>(sW0 + W0B + r * GB_P0 + c4 * 2) = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), bf16x4);
    }
    for (int i = tid; i < 32 * 32; i += 256) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        const f32x4 x = *reinterpret_cast<const f32x4*>(p.w3 + r * 128 + c4);
        const bf16x4 h = cv4(x);
        *reinterpret_cast<bf16x4*>(sW3 + r * GB_P3 + c4 * 2) = h;
        if (PL == 2) *reinterpret_cast<bf16x4*>(sW3 + W3B + r * GB_P3 + c4 * 2) = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), bf16x4);
    }
    float b3f[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b3f[r] = p.b3[crow32(r, hi)];
    __syncthreads();

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // Work unit = 32 consecutive edges x 4 heads: a wave's 32 rows are 32 EDGES of ONE head (head = 4 (unit & 1) + wave).
    // Edge lists are source-major (reference dataset_3dssg.py:264-266), so the 32 lanes of a Gq load mostly name the same
    // node row: one cache line per instruction instead of 32 (row_map = 0: the older 4 edges x 8 heads per wave).
    const int n_units = 2 * ((p.n_edges + 31) / 32);
    for (int g = blockIdx.x; g < n_units; g += gridDim.x) {
        asm volatile("" ::: "memory");                    // keep the weight fragments out of LICM's hands (edge_gate.hip)
        const int e_raw = p.row_map ? (g >> 1) * 32 + li : g * 16 + wave * 4 + (li >> 3);
        const int h = p.row_map ? (g & 1) * 4 + wave : li & 7;
        const bool valid = e_raw < p.n_edges;
        const int e = valid ? e_raw : p.n_edges - 1;
        // ---- this row's kproj values: k-slot (hi, e) of step ks is c = 16 ks + 8 hi + e ----
        bf16x8 zh[4], zl[4];
        if (p.use_edge) {
            const float* zrow = p.kproj + (size_t)e * 512 + h * 64 + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (KS >= 2) {                            // eight bf16 (fp16) = one 16-byte load
                    zh[ks] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(p.kproj + (size_t)e * 512) + (h * 64 + 8 * hi + 16 * ks) * 2);
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
        const float* gq = p.node + (size_t)sn * p.ld_node + p.gq_off + h * 128 + 4 * hi;
        f32x16 lg;
#pragma unroll
        for (int r = 0; r < 16; ++r) lg[r] = b3f[r];
#pragma unroll
        for (int to = 0; to < 4; ++to) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (p.use_edge) {                             // (USE_GCN_EDGE=false: hidden = relu(Gq), the edge half is absent)
                const char* ap = sW0 + (to * 32 + li) * GB_P0 + 16 * hi;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + 32 * ks);
                    if (PL == 2) {
                        const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + W0B + 32 * ks);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, zh[ks], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, zl[ks], acc, 0, 0, 0);
                    }
                    acc = mfma_h<F16>(ah, zh[ks], acc);
                }
            }
            // hidden = relu(acc + Gq[src, h*128 + o]),  o = to*32 + 8*r4 + 4*hi + c  (registers r = 4 r4 + c)
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
                // W3[m = li][o = to*32 + 16 half + 4 hi + {0..3}] and the same + 8
                const char* wp = sW3 + li * GB_P3 + (to * 32 + 16 * half + 4 * hi) * 2;
                const bf16x4 wa = *reinterpret_cast<const bf16x4*>(wp), wb = *reinterpret_cast<const bf16x4*>(wp + 16);
                const bf16x8 wh = __builtin_shufflevector(wa, wb, 0, 1, 2, 3, 4, 5, 6, 7);
                if (PL == 2) {
                    const bf16x4 l0 = __builtin_convertvector(p0 - __builtin_convertvector(h0, f32x4), bf16x4);
                    const bf16x4 l1 = __builtin_convertvector(p1 - __builtin_convertvector(h1, f32x4), bf16x4);
                    const bf16x8 hl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x4 la = *reinterpret_cast<const bf16x4*>(wp + W3B), lb = *reinterpret_cast<const bf16x4*>(wp + W3B + 16);
                    const bf16x8 wl = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
                    lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, hh, lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, hl, lg, 0, 0, 0);
                }
                lg = mfma_h<F16>(wh, hh, lg);
            }
        }
        // softmax over the 32 channels m = crow32(r, hi) (+ the other 16 in lane^32), times value
        float mx = lg[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, lg[r]);
        mx = half_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            lg[r] = __expf(lg[r] - mx);
            sum += lg[r];
        }
        sum = half_sum(sum);
        const float inv = 1.f / sum;
        if (p.agg) {                       // fused max aggregation: the gated rows are never stored (gate_agg.h)
            gate_aggregate_max(smem + PL * (W0B + W3B) + wave * AG_WAVE_BYTES, lg, inv, p.node + (size_t)dn * p.ld_node + p.v_off + h * 32 + 4 * hi,
                               valid ? sn : -1, li, hi, lane, h, p.agg, p.ld_agg);
        } else if (valid) {
            const float* vrow = p.node + (size_t)dn * p.ld_node + p.v_off + h * 32 + 4 * hi;
            float* grow = p.gated + (size_t)e * 256 + h * 32 + 4 * hi;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(vrow + 8 * r4);
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = lg[r4 * 4 + c] * inv * v[c];
                *reinterpret_cast<f32x4*>(grow + 8 * r4) = o;
            }
            if (p.prob) {
#pragma unroll
                for (int r = 0; r < 16; ++r) p.prob[(size_t)e * 256 + crow32(r, hi) * 8 + h] = lg[r] * inv;
            }
        }
    }
}

}  // namespace

int launch_edge_gate_bf16(const GateArgs& a, int terms, int kproj_split, hipStream_t s, const GateArgs* twin) {
    if (a.n_edges <= 0) return 0;
    if (twin && (twin->n_edges != a.n_edges || !twin->agg != !a.agg || twin->row_map != a.row_map || twin->use_edge != a.use_edge ||
                 !twin->prob != !a.prob || twin->grid_cap != a.grid_cap || twin->src != a.src || twin->dst != a.dst))
        return fail(-1, "edge_gate_bf16: a twin launch needs two problems on the same edge list with the same options");
    const GateArgs& b = twin ? *twin : a;
    if ((a.ld_node & 3) || (a.gq_off & 3) || (a.v_off & 3)) return fail(-1, "edge_gate: ld_node/gq_off/v_off must be multiples of 4");
    if (terms != 1 && terms != 3) return fail(-1, "edge_gate_bf16: terms must be 1 or 3");
    if (a.agg && (!a.row_map || a.prob || (a.ld_agg & 3))) return fail(-1, "edge_gate_bf16: the fused aggregation needs the 32-edges-per-wave row map and no prob tap");
    const int n_groups = a.row_map ? 2 * ((a.n_edges + 31) / 32) : (a.n_edges + 15) / 16;
    // persistent grid (weights staged once per block): three blocks per CU; the single-rounding kernel holds four (128 VGPRs,
    // 37 KB of LDS) and 1024 measured 0.7 % faster per step than 768 or 1280 with the aggregation fused in
    const int cap = a.grid_cap > 0 ? a.grid_cap : terms == 1 ? 1024 : 768;
    const int grid = n_groups < cap ? n_groups : cap;
#define VLSAT_GB(T, K) do { if (twin) hipLaunchKernelGGL((edge_gate_bf16_kernel<T, K, true>), dim3(grid, 2), dim3(256), 0, s, a, b); \
                            else hipLaunchKernelGGL((edge_gate_bf16_kernel<T, K, false>), dim3(grid), dim3(256), 0, s, a, a); } while (0)
    if (kproj_split >= 2 && terms != 1) return fail(-1, "edge_gate_bf16: half-row kproj needs terms = 1");
    if (terms == 3) { if (kproj_split) VLSAT_GB(3, 1); else VLSAT_GB(3, 0); }
    else            { if (kproj_split == 3) VLSAT_GB(1, 3); else if (kproj_split == 2) VLSAT_GB(1, 2); else if (kproj_split) VLSAT_GB(1, 1); else VLSAT_GB(1, 0); }       // (3: fp16 half rows)
#undef VLSAT_GB
    VLSAT_LAUNCH_CHECK("edge_gate_bf16");
    return 0;
}

}  // namespace vlsat
