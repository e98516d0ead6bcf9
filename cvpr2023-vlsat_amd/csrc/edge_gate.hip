// 'fat' edge gate of MultiHeadedEdgeAttention.forward (reference network_MMG.py:96-104):
//   q = proj_query(x_i).view(E,64,8); k = proj_edge(e).view(E,64,8)          (head = FAST axis)
//   prob = softmax_dim1( Conv1d(128->32)( ReLU( Conv1d(128->128)( cat[q,k] ) ) ) )   [E,32,8]
//   gated = prob.reshape(E,256) * proj_value(x_j)
//
// Algebra used here (weights prepared once in vlsat_finalize_weights):
//   * the q half of layer 1 only depends on the SOURCE node -> Gq[n, h*128+o] (bias included)
//     is computed per node by the node-side GEMM and gathered here;
//   * proj_edge rows are permuted so k arrives head-major: kproj[e, h*64 + c] == k[e, c, h];
//     the [E,512] matrix is then a contiguous [8E, 64] matrix of (edge, head) rows.
// Per (edge, head) row:  hidden = relu(Gq + W0k . kproj_row);  logits = W3 . hidden + b3;
// prob = softmax(logits);  gated[e, h*32+m] = prob[m] * value[dst[e], h*32+m]   (value and gated are kept
// HEAD-MAJOR -- the engine permutes proj_value's rows and prop.0's columns once -- so the four consecutive
// channels m a lane owns per MFMA row group are one float4 load and one float4 store).
//
// fp32 MFMA, transposed products so that a lane owns ONE (edge, head) row:
//   hidden^T[o][row] : A = W0k (LDS), B = kproj rows straight from HBM (float4 per lane)
//   logits^T[m][row] : A = W3 (LDS), B = hidden^T registers of layer 1 (no LDS round trip)
// so the softmax over the 32 channels is 15 in-lane ops + one lane^32 exchange.
// One wave = 32 rows = 4 edges per step (192 MFMAs); a block of 4 waves walks 16-edge groups.
#include <cstdlib>
#include "gemm_core.h"
#include "gate_agg.h"
#include "kernels.h"

namespace vlsat {

constexpr int GT_PITCH = 68;     // W0k rows: 64 + 4 pad
constexpr int GT_PITCH3 = 132;   // W3 rows: 128 + 4 pad

// FUSED: the max aggregation runs inside the kernel (GateArgs::agg; gate_agg.h) and its wave buffers exist -- 63 KB of LDS per block,
// two blocks per CU.  Exact fp32 keeps the separate aggregate launch by default, and that variant holds 51.7 KB: three per CU.
// TWIN (round 6): the gates of gcn_3ds[l] and gcn_2ds[l] of a one-scene plan in ONE launch, selected by blockIdx.y (same edge list,
// same grid; every block runs the single launch's code on its own problem: bit-identical results).
template <bool FUSED, bool TWIN = false>
__global__ __launch_bounds__(256, 2) void edge_gate_kernel(GateArgs pa, GateArgs pb) {
    const GateArgs& p = (TWIN && blockIdx.y != 0) ? pb : pa;
    __shared__ __attribute__((aligned(16))) float sW0[128 * GT_PITCH];
    __shared__ __attribute__((aligned(16))) float sW3[32 * GT_PITCH3];
    __shared__ __attribute__((aligned(16))) char sAgg[FUSED ? 4 * AG_WAVE_BYTES : 16];      // wave buffers of the fused aggregation
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;

    for (int i = tid; i < 128 * 16; i += 256) {
        const int r = i >> 4, c4 = (i & 15) * 4;
        *reinterpret_cast<f32x4*>(sW0 + r * GT_PITCH + c4) = *reinterpret_cast<const f32x4*>(p.w0k + r * 64 + c4);
    }
    for (int i = tid; i < 32 * 32; i += 256) {
        const int r = i >> 5, c4 = (i & 31) * 4;
        *reinterpret_cast<f32x4*>(sW3 + r * GT_PITCH3 + c4) = *reinterpret_cast<const f32x4*>(p.w3 + r * 128 + c4);
    }
    float b3f[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b3f[r] = p.b3[crow32(r, hi)];
    __syncthreads();

    // Work unit = 32 consecutive edges x 4 heads: a wave's 32 rows are 32 EDGES of ONE head (head = 4 (unit & 1) + wave).
    // Edge lists are source-major (reference dataset_3dssg.py:264-266), so the 32 lanes of a Gq load mostly name the same
    // node row: one cache line per instruction instead of 32 (row_map = 0: the older 4 edges x 8 heads per wave).
    const int n_units = 2 * ((p.n_edges + 31) / 32);
    for (int g = blockIdx.x; g < n_units; g += gridDim.x) {
        // the weight fragments are loop-invariant; without this clobber LICM hoists all 192 of
        // them into VGPRs and the kernel spills.  Re-reading them from LDS per step is free.
        asm volatile("" ::: "memory");
        const int e_raw = p.row_map ? (g >> 1) * 32 + li : g * 16 + wave * 4 + (li >> 3);
        const int h = p.row_map ? (g & 1) * 4 + wave : li & 7;
        const bool valid = e_raw < p.n_edges;
        const int e = valid ? e_raw : p.n_edges - 1;
        const float* zrow = p.kproj + (size_t)e * 512 + h * 64 + 4 * hi;
        f32x4 z[8];
        if (p.use_edge) {
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) z[kg] = *reinterpret_cast<const f32x4*>(zrow + kg * 8);
        } else {                                   // USE_GCN_EDGE=false: hidden = relu(Gq), the edge half is absent
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) z[kg] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int sn = p.src[e], dn = p.dst[e];

        // Per 32-wide slice `to` of the hidden layer: layer 1 (32 MFMAs) then immediately its
        // contribution to layer 2 (16 MFMAs), so only one 32x32 accumulator is live at a time.
        const float* gq = p.node + (size_t)sn * p.ld_node + p.gq_off + h * 128 + 4 * hi;
        f32x16 lg;
#pragma unroll
        for (int r = 0; r < 16; ++r) lg[r] = b3f[r];
#pragma unroll
        for (int to = 0; to < 4; ++to) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sW0 + (to * 32 + li) * GT_PITCH + kg * 8 + 4 * hi);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], z[kg][s], acc, 0, 0, 0);
            }
            // hidden = relu(acc + Gq[src, h*128 + o]),  o = to*32 + 8*r4 + 4*hi + c
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 gqv = *reinterpret_cast<const f32x4*>(gq + to * 32 + 8 * r4);
                // layer-2 A fragment: W3[m = li][o = to*32 + 8*r4 + 4*hi + c]
                const f32x4 w3v = *reinterpret_cast<const f32x4*>(sW3 + li * GT_PITCH3 + to * 32 + 8 * r4 + 4 * hi);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float hid = fmaxf(acc[r4 * 4 + c] + gqv[c], 0.f);
                    lg = __builtin_amdgcn_mfma_f32_32x32x2f32(w3v[c], hid, lg, 0, 0, 0);
                }
            }
        }
        // softmax over the 32 channels m = crow32(r, hi) (+ the other 16 in lane^32)
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
        if (FUSED && p.agg) {                  // fused max aggregation: the gated rows are never stored (gate_agg.h)
            gate_aggregate_max(sAgg + wave * AG_WAVE_BYTES, lg, inv, p.node + (size_t)dn * p.ld_node + p.v_off + h * 32 + 4 * hi, valid ? sn : -1,
                               li, hi, lane, h, p.agg, p.ld_agg);
        } else if (valid) {
            // lane's channels: m = 8*r4 + 4*hi + c  (crow32), c = 0..3 -> one float4 per r4
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
            if (p.prob) {                      // test tap in the reference's [E, 32, 8] order
#pragma unroll
                for (int r = 0; r < 16; ++r) p.prob[(size_t)e * 256 + crow32(r, hi) * 8 + h] = lg[r] * inv;
            }
        }
    }
}

// Generic head geometry (MODEL.NUM_HEADS / DIM_ATTEN other than 8 / 256; reference network_MMG.py:48-50): d_k = 512 / H
// query / edge channels per head, hidden width 2 d_k, d_o = DIM_ATTEN / H output channels.  Same algebra and layouts as
// above (Gq per node, head-major kproj / value / gated), plain VALU: one thread per (edge, head), the hidden vector in
// LDS (transposed, so a warp's accesses are conflict-free), weights read with wave-uniform (scalar) loads.  The MFMA
// kernels above are built for the shipped 8 x (64, 64, 32) only.
__global__ __launch_bounds__(64) void edge_gate_generic_kernel(GateArgs p, int n_heads, int dk, int dox) {
    extern __shared__ float hid[];                      // [2 dk][64]
    const int lane = threadIdx.x;
    const long row = (long)blockIdx.x * 64 + lane;      // (edge, head)
    const long n_rows = (long)p.n_edges * n_heads;
    const bool valid = row < n_rows;
    const long rr = valid ? row : n_rows - 1;
    const int e = (int)(rr / n_heads), h = (int)(rr % n_heads);
    const int HID = 2 * dk;
    const float* z = p.kproj + (size_t)e * (n_heads * dk) + h * dk;
    const float* gq = p.node + (size_t)p.src[e] * p.ld_node + p.gq_off + h * HID;
    for (int o = 0; o < HID; ++o) {
        float a = gq[o];
        if (p.use_edge)
            for (int c = 0; c < dk; ++c) a = fmaf(p.w0k[o * dk + c], z[c], a);
        hid[o * 64 + lane] = fmaxf(a, 0.f);
    }
    float mx = -INFINITY;
    for (int m = 0; m < dox; ++m) {                     // pass 1: maximum of the logits
        float a = p.b3[m];
        for (int o = 0; o < HID; ++o) a = fmaf(p.w3[m * HID + o], hid[o * 64 + lane], a);
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
    float* grow = p.gated + (size_t)e * (n_heads * dox) + h * dox;
    const float* vrow = p.node + (size_t)p.dst[e] * p.ld_node + p.v_off + h * dox;
    for (int m = 0; m < dox; ++m) {                     // pass 2: exponentials (kept in the output row), their sum
        float a = p.b3[m];
        for (int o = 0; o < HID; ++o) a = fmaf(p.w3[m * HID + o], hid[o * 64 + lane], a);
        a = __expf(a - mx);
        sum += a;
        if (valid) grow[m] = a;
    }
    if (!valid) return;
    const float inv = 1.f / sum;
    for (int m = 0; m < dox; ++m) {
        const float pr = grow[m] * inv;
        if (p.prob) p.prob[(size_t)e * (n_heads * dox) + m * n_heads + h] = pr;
        grow[m] = pr * vrow[m];
    }
}

int launch_edge_gate_generic(const GateArgs& a, int n_heads, int dk, int dox, hipStream_t s) {
    if (a.n_edges <= 0) return 0;
    if (dk < 1 || dk > 128 || dox < 1) return fail(-1, "edge_gate: unsupported head geometry");
    const long rows = (long)a.n_edges * n_heads;
    hipLaunchKernelGGL(edge_gate_generic_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 2 * dk * 64 * sizeof(float), s, a,
                       n_heads, dk, dox);
    VLSAT_LAUNCH_CHECK("edge_gate_generic");
    return 0;
}

int launch_edge_gate(const GateArgs& a, hipStream_t s, const GateArgs* twin) {
    if (a.n_edges <= 0) return 0;
    if (twin && (twin->n_edges != a.n_edges || !twin->agg != !a.agg || twin->row_map != a.row_map || twin->use_edge != a.use_edge ||
                 !twin->prob != !a.prob || twin->grid_cap != a.grid_cap || twin->src != a.src || twin->dst != a.dst))
        return fail(-1, "edge_gate: a twin launch needs two problems on the same edge list with the same options");
    if (a.agg && (!a.row_map || a.prob || (a.ld_agg & 3))) return fail(-1, "edge_gate: the fused aggregation needs the 32-edges-per-wave row map and no prob tap");
    if ((a.ld_node & 3) || (a.gq_off & 3) || (a.v_off & 3)) return fail(-1, "edge_gate: ld_node/gq_off/v_off must be multiples of 4");
    const int n_groups = a.row_map ? 2 * ((a.n_edges + 31) / 32) : (a.n_edges + 15) / 16;
    // persistent: every block stages the weights once and walks ~n_groups / grid groups; the grid is what is resident at once -- 3
    // blocks per CU (51.7 KB of LDS each), 2 with the aggregation's wave buffers (63 KB) -- so there is no partial last wave of blocks
    const int cap = a.grid_cap > 0 ? a.grid_cap : (a.agg ? 512 : 768);
    const int grid = n_groups < cap ? n_groups : cap;
    if (twin) {
        if (a.agg) hipLaunchKernelGGL((edge_gate_kernel<true, true>), dim3(grid, 2), dim3(256), 0, s, a, *twin);
        else hipLaunchKernelGGL((edge_gate_kernel<false, true>), dim3(grid, 2), dim3(256), 0, s, a, *twin);
    } else if (a.agg) hipLaunchKernelGGL((edge_gate_kernel<true, false>), dim3(grid), dim3(256), 0, s, a, a);
    else hipLaunchKernelGGL((edge_gate_kernel<false, false>), dim3(grid), dim3(256), 0, s, a, a);
    VLSAT_LAUNCH_CHECK("edge_gate");
    return 0;
}

}  // namespace vlsat
