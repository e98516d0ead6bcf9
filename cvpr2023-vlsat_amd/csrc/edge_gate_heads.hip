// 'fat' edge gate on the fp32 matrix cores for the head geometries other than the shipped 8 x (64, 64, 32):
// MODEL.NUM_HEADS in {4, 8, 16} and DIM_ATTEN in {128, 256, 512} (reference network_MMG.py:48-50,69-79,96-104) give
// d_k = 512 / H query / edge channels per head in {128, 64, 32}, a hidden layer of 2 d_k and d_o = DIM_ATTEN / H output
// channels in {8 .. 128}.  Same algebra, data flow and lane model as edge_gate.hip:
//   per (edge, head) row:  hidden = relu(Gq[src] + W0k . kproj_row),  logits = W3 . hidden + b3,
//                          prob = softmax over the d_o channels,  gated = prob * value[dst]   (head-major)
//   a wave owns 32 consecutive EDGES of one head (source-major edge lists: its Gq loads name one or two node rows);
//   both layers as transposed v_mfma_f32_32x32x2_f32 products so that a lane owns one row and the softmax stays in-lane;
//   the hidden layer goes 32 outputs at a time from the layer-1 accumulator straight into the layer-2 product.
// What the template adds: TO = 2 d_k / 32 hidden slices, d_k / 2 MFMAs per slice, MO = ceil(d_o / 32) logit blocks (rows of
// W3 past d_o are zero, their channels masked out of the softmax).  Weights: W0k [2 d_k][d_k] and W3 [d_o][2 d_k] in LDS; at
// d_k = 128 W0k alone is 135 KB, so there the block is 8 waves (one block per CU) and W3's fragments come from global
// memory (64 .. 128 KB, L2-resident; 25 GB/s per CU against the 49 000 MFMA cycles of a wave step).
// Before this kernel these geometries ran on the VALU kernel of edge_gate.hip (kept for anything else): 266 ms per bench
// step at 4 heads, 77 ms at 16, against 1.1 ms at 8.
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

template <int DK, int DOX>
__global__ __launch_bounds__(DK == 128 ? 512 : 256, DK == 128 ? 1 : 2) void edge_gate_hd_kernel(GateArgs p, int n_heads) {
    constexpr int HID = 2 * DK, TO = HID / 32, MO = (DOX + 31) / 32, KG = DK / 8;
    constexpr int P0 = DK + 4, P3 = HID + 4;                 // LDS row pitches (floats): 16-byte pad, conflict-free b128 reads
    constexpr bool W3_LDS = DK != 128;
    constexpr bool Z_REGS = !(DK == 128 && DOX > 32);        // else: the row's kproj values are re-read per hidden slice (L1 hits) --
                                                             // 64 of them next to 32 .. 64 logit accumulators do not fit the registers
    constexpr int NT = DK == 128 ? 512 : 256, NW = NT / 64;
    static_assert(DOX % 8 == 0 && (W3_LDS || DOX % 32 == 0), "output channels per head");
    __shared__ __attribute__((aligned(16))) float sW0[HID * P0];
    __shared__ __attribute__((aligned(16))) float sW3[W3_LDS ? MO * 32 * P3 : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int A = n_heads * DOX;                              // row width of gated / prob / the value columns

    for (int i = tid; i < HID * (DK / 4); i += NT) {
        const int r = i / (DK / 4), c4 = (i % (DK / 4)) * 4;
        *reinterpret_cast<f32x4*>(sW0 + r * P0 + c4) = *reinterpret_cast<const f32x4*>(p.w0k + r * DK + c4);
    }
    if (W3_LDS) {
        for (int i = tid; i < MO * 32 * (HID / 4); i += NT) {
            const int r = i / (HID / 4), c4 = (i % (HID / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < DOX) v = *reinterpret_cast<const f32x4*>(p.w3 + r * HID + c4);
            *reinterpret_cast<f32x4*>(sW3 + r * P3 + c4) = v;
        }
    }
    __syncthreads();

    const long n_wu = (long)((p.n_edges + 31) / 32) * n_heads;          // wave units: (block of 32 edges, head)
    for (long u = blockIdx.x; u * NW < n_wu; u += gridDim.x) {
        asm volatile("" ::: "memory");                        // keep the weight fragments out of LICM's hands (edge_gate.hip)
        const long wu = u * NW + wave;
        if (wu >= n_wu) continue;                             // (no barrier in this loop)
        const int h = (int)(wu % n_heads);
        const int e_raw = (int)(wu / n_heads) * 32 + li;
        const bool valid = e_raw < p.n_edges;
        const int e = valid ? e_raw : p.n_edges - 1;
        const float* zrow = p.kproj + (size_t)e * 512 + h * DK + 4 * hi;
        f32x4 z[Z_REGS ? KG : 1];
        if (Z_REGS && p.use_edge) {
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) z[kg] = *reinterpret_cast<const f32x4*>(zrow + kg * 8);
        }
        const int sn = p.src[e], dn = p.dst[e];
        const float* gq = p.node + (size_t)sn * p.ld_node + p.gq_off + h * HID + 4 * hi;
        f32x16 lg[MO];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {                  // b3[m], m = mo*32 + 8*r4 + 4*hi + c (zero past d_o)
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (mo * 32 + 8 * r4 < DOX) b = *reinterpret_cast<const f32x4*>(p.b3 + mo * 32 + 8 * r4 + 4 * hi);
#pragma unroll
                for (int c = 0; c < 4; ++c) lg[mo][r4 * 4 + c] = b[c];
            }
#pragma unroll(DK == 128 ? 1 : TO)                        // (d_k = 128: one slice is 64 + 64 MFMAs; unrolled, the scheduler's appetite spills)
        for (int to = 0; to < TO; ++to) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (p.use_edge) {                                 // (USE_GCN_EDGE=false: hidden = relu(Gq), the edge half is absent)
#pragma unroll
                for (int kg = 0; kg < KG; ++kg) {
                    if (!Z_REGS && (kg & 3) == 0) asm volatile("" ::: "memory");     // (at most four re-read groups in flight)
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sW0 + (to * 32 + li) * P0 + kg * 8 + 4 * hi);
                    const f32x4 zk = Z_REGS ? z[Z_REGS ? kg : 0] : *reinterpret_cast<const f32x4*>(zrow + kg * 8);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], zk[s], acc, 0, 0, 0);
                }
            }
            // hidden = relu(acc + Gq[src, h*HID + o]),  o = to*32 + 8*r4 + 4*hi + c; layer-2 A fragment: W3[m][o]
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (!W3_LDS) asm volatile("" ::: "memory");                           // (W3 fragments from global: one r4 group at a time)
                const f32x4 gqv = *reinterpret_cast<const f32x4*>(gq + to * 32 + 8 * r4);
                float hid[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) hid[c] = fmaxf(acc[r4 * 4 + c] + gqv[c], 0.f);
#pragma unroll
                for (int mo = 0; mo < MO; ++mo) {
                    const f32x4 w3v = W3_LDS ? *reinterpret_cast<const f32x4*>(sW3 + (mo * 32 + li) * P3 + to * 32 + 8 * r4 + 4 * hi)
                                             : *reinterpret_cast<const f32x4*>(p.w3 + (size_t)(mo * 32 + li) * HID + to * 32 + 8 * r4 + 4 * hi);
#pragma unroll
                    for (int c = 0; c < 4; ++c) lg[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(w3v[c], hid[c], lg[mo], 0, 0, 0);
                }
            }
        }
        // softmax over the d_o channels m = mo*32 + crow32(r, hi) (+ the other 16 of a block in lane^32)
        float mx = -INFINITY;
#pragma unroll
        for (int mo = 0; mo < MO; ++mo)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mo * 32 + 8 * (r >> 2) < DOX) mx = fmaxf(mx, lg[mo][r]);       // (a group of four channels is in or out as a whole)
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

template <int DK, int DOX>
int run(const GateArgs& a, int n_heads, hipStream_t s) {
    constexpr int NT = DK == 128 ? 512 : 256, NW = NT / 64;
    const long n_wu = (long)((a.n_edges + 31) / 32) * n_heads, units = (n_wu + NW - 1) / NW;
    const long cap = a.grid_cap > 0 ? a.grid_cap : (DK == 128 ? 256 : 768);      // persistent: weights staged once per block
    hipLaunchKernelGGL((edge_gate_hd_kernel<DK, DOX>), dim3((unsigned)std::min(units, cap)), dim3(NT), 0, s, a, n_heads);
    return 0;
}

}  // namespace

// 1 = geometry not built here (the caller falls back to the VALU kernel)
int launch_edge_gate_heads(const GateArgs& a, int n_heads, int dk, int dox, hipStream_t s) {
    if (a.n_edges <= 0) return 0;
    if ((a.ld_node & 3) || (a.gq_off & 3) || (a.v_off & 3)) return fail(-1, "edge_gate: ld_node/gq_off/v_off must be multiples of 4");
    if (n_heads * dk != 512) return 1;
    int r = 1;
#define VLSAT_GH(DK, DOX) if (dk == DK && dox == DOX) r = run<DK, DOX>(a, n_heads, s)
    VLSAT_GH(32, 8); VLSAT_GH(32, 16); VLSAT_GH(32, 32);
    VLSAT_GH(64, 16); VLSAT_GH(64, 32); VLSAT_GH(64, 64);
    VLSAT_GH(128, 32); VLSAT_GH(128, 64); VLSAT_GH(128, 128);
#undef VLSAT_GH
    if (r) return r;
    VLSAT_LAUNCH_CHECK("edge_gate_heads");
    return 0;
}

}  // namespace vlsat
