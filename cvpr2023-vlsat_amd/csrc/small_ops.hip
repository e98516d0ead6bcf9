// Small fused VALU kernels of the path (all HBM/latency-bound glue around the MFMA kernels).
#include "common.h"
#include "kernels.h"

namespace vlsat {

// ------------------------------------------------------------------------------------------
// Edge descriptor + first layer of both relation encoders.
//   ed = [x_i[0:6] - x_j[0:6], log(x_i[6:11] / x_j[6:11])]   with x_i = desc[src], x_j = desc[dst]
//   (Gen_edge_descriptor.message, flow='target_to_source': reference src/utils/op_utils.py:78-97)
//   h1[e, c] = relu(W1cat[c, :] . ed + b1cat[c]),  c < 64: rel_encoder_3d.conv1, c >= 64: rel_encoder_2d.conv1
//   (PointNetfeat conv1 with point_size = 11, P = 1: reference network_PointNet.py:141-144)
// Block = 64 edges: 64 threads build the descriptors in LDS, then 256 threads x 32 outputs.
__global__ __launch_bounds__(256) void edge_embed_kernel(const float* __restrict__ desc,
                                                         const int32_t* __restrict__ src,
                                                         const int32_t* __restrict__ dst, int n_edges,
                                                         const float* __restrict__ w1cat,
                                                         const float* __restrict__ b1cat, float* __restrict__ h1) {
    __shared__ float sE[64][12];
    __shared__ float sWt[11][129];     // transposed weights, padded
    __shared__ float sB[128];
    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * 64;
    for (int i = tid; i < 128 * 11; i += 256) sWt[i % 11][i / 11] = w1cat[i];
    if (tid < 128) sB[tid] = b1cat[tid];
    if (tid < 64) {
        int e = e0 + tid;
        e = e < n_edges ? e : n_edges - 1;
        const float* a = desc + (size_t)src[e] * 11;
        const float* b = desc + (size_t)dst[e] * 11;
#pragma unroll
        for (int c = 0; c < 6; ++c) sE[tid][c] = a[c] - b[c];
#pragma unroll
        for (int c = 6; c < 11; ++c) sE[tid][c] = logf(a[c] / b[c]);
    }
    __syncthreads();
    const int c = tid & 127;
    float w[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) w[k] = sWt[k][c];
    const float bb = sB[c];
    for (int el = tid >> 7; el < 64; el += 2) {
        const int e = e0 + el;
        if (e >= n_edges) break;
        float acc = bb;
#pragma unroll
        for (int k = 0; k < 11; ++k) acc = fmaf(w[k], sE[el][k], acc);
        h1[(size_t)e * 128 + c] = fmaxf(acc, 0.f);
    }
}

int launch_edge_embed(const float* desc, const int32_t* src, const int32_t* dst, int n_edges,
                      const float* w1cat, const float* b1cat, float* h1, hipStream_t s) {
    if (n_edges <= 0) return 0;
    hipLaunchKernelGGL(edge_embed_kernel, dim3((n_edges + 63) / 64), dim3(256), 0, s, desc, src, dst, n_edges,
                       w1cat, b1cat, h1);
    VLSAT_LAUNCH_CHECK("edge_embed");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Spatial tail of the 3D node feature: x[n, col0:col0+8] = [desc[3:9], log desc[9], log desc[10]]
// (reference SGFN_MMG/model.py:296-299).
__global__ void desc_tail_kernel(const float* __restrict__ desc, int n_nodes, float* __restrict__ x, int ldx,
                                 int col0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes * 8) return;
    const int n = i >> 3, c = i & 7;
    float v = desc[(size_t)n * 11 + 3 + c];
    if (c >= 6) v = logf(v);
    x[(size_t)n * ldx + col0 + c] = v;
}
int launch_desc_tail(const float* desc, int n_nodes, float* x, int ldx, int col0, hipStream_t s) {
    if (n_nodes <= 0) return 0;
    hipLaunchKernelGGL(desc_tail_kernel, dim3((n_nodes * 8 + 255) / 256), dim3(256), 0, s, desc, n_nodes, x, ldx, col0);
    VLSAT_LAUNCH_CHECK("desc_tail");
    return 0;
}

// ------------------------------------------------------------------------------------------
// In-place LayerNorm over rows of 512 (eps 1e-5, biased variance = torch.nn.LayerNorm), optional
// ReLU.  One wave per row, 8 values per lane as two float4 (columns 4*lane and 256 + 4*lane).
// reference transformer/attention.py:122 (post-LN residual) and network_MMG.py:236-248 (ReLU).
// Optional residual (rows of `resid`; r_split 1: split-pair words, 2: half rows): y = LN(x + resid) -- the post-LN residual of the edge
// attention in the split-bf16 mode, where adding it here is cheaper than as an accumulator init of the out-projection.
__global__ __launch_bounds__(256) void layernorm512_kernel(const float* __restrict__ x, int ld, float* __restrict__ y, int ldy,
                                                           int rows, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int relu, int out_split,
                                                           const float* __restrict__ resid, int ldr, int r_split, int x_f16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + (size_t)row * ld;
    f32x4 a, b;
    if (x_f16) {                                           // fp16 half rows: element n at byte 2 n
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        a = __builtin_convertvector(reinterpret_cast<const f16x4_t*>(p)[lane], f32x4);
        b = __builtin_convertvector(reinterpret_cast<const f16x4_t*>(p)[64 + lane], f32x4);
    } else {
        a = *reinterpret_cast<const f32x4*>(p + 4 * lane);
        b = *reinterpret_cast<const f32x4*>(p + 256 + 4 * lane);
    }
    if (resid) {
        const float* r = resid + (size_t)row * ldr;
        f32x4 ra, rb;
        if (r_split == 4) {                                // half rows of fp16 (precision mode fp16_mixed)
            typedef _Float16 f16x4_r __attribute__((ext_vector_type(4)));
            ra = __builtin_convertvector(reinterpret_cast<const f16x4_r*>(r)[lane], f32x4);
            rb = __builtin_convertvector(reinterpret_cast<const f16x4_r*>(r)[64 + lane], f32x4);
        } else if (r_split == 2) {                         // half rows: bf16 at byte 2 * column
            const uint2 ha = reinterpret_cast<const uint2*>(r)[lane], hb = reinterpret_cast<const uint2*>(r)[64 + lane];
            ra = f32x4{__uint_as_float(ha.x << 16), __uint_as_float(ha.x & 0xffff0000u), __uint_as_float(ha.y << 16), __uint_as_float(ha.y & 0xffff0000u)};
            rb = f32x4{__uint_as_float(hb.x << 16), __uint_as_float(hb.x & 0xffff0000u), __uint_as_float(hb.y << 16), __uint_as_float(hb.y & 0xffff0000u)};
        } else {
            ra = *reinterpret_cast<const f32x4*>(r + 4 * lane);
            rb = *reinterpret_cast<const f32x4*>(r + 256 + 4 * lane);
            if (r_split) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { ra[c] = unpack_split(ra[c]); rb[c] = unpack_split(rb[c]); }
            }
        }
        a += ra;
        b += rb;
    }
    float s = a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2] + b[3];
    const float mean = wave_sum(s) * (1.f / 512.f);
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[c] -= mean; b[c] -= mean;
        v += a[c] * a[c] + b[c] * b[c];
    }
    const float rstd = rsqrtf(wave_sum(v) * (1.f / 512.f) + 1e-5f);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + 4 * lane);
    const f32x4 gb = *reinterpret_cast<const f32x4*>(gamma + 256 + 4 * lane);
    const f32x4 ba = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + 256 + 4 * lane);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[c] = a[c] * rstd * ga[c] + ba[c];
        b[c] = b[c] * rstd * gb[c] + bb[c];
        if (relu) { a[c] = fmaxf(a[c], 0.f); b[c] = fmaxf(b[c], 0.f); }
        if (out_split == 1) { a[c] = pack_split(a[c]); b[c] = pack_split(b[c]); }     // bf16 modes: the consumers are GEMM A operands
    }
    float* q = y + (size_t)row * ldy;
    if (out_split == 4) {                                  // half rows of fp16 (clamped: the consumers are fp16 MFMA operands)
        typedef _Float16 f16x4_o __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[c] = __builtin_amdgcn_fmed3f(a[c], -65504.f, 65504.f); b[c] = __builtin_amdgcn_fmed3f(b[c], -65504.f, 65504.f); }
        f16x4_o* h = reinterpret_cast<f16x4_o*>(q);
        h[lane] = __builtin_convertvector(a, f16x4_o);
        h[64 + lane] = __builtin_convertvector(b, f16x4_o);
        return;
    }
    if (out_split == 2) {                                  // half rows: bf16 at byte 2 * column
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        bf16x4_t* h = reinterpret_cast<bf16x4_t*>(q);
        h[lane] = __builtin_convertvector(a, bf16x4_t);
        h[64 + lane] = __builtin_convertvector(b, bf16x4_t);
        return;
    }
    *reinterpret_cast<f32x4*>(q + 4 * lane) = a;
    *reinterpret_cast<f32x4*>(q + 256 + 4 * lane) = b;
}
int launch_layernorm(float* x, int ld, int rows, int dim, const float* gamma, const float* beta, int relu,
                     hipStream_t s) {
    return launch_layernorm_to(x, ld, x, ld, rows, dim, gamma, beta, relu, 0, s);
}
int launch_layernorm_to(const float* x, int ld, float* y, int ldy, int rows, int dim, const float* gamma, const float* beta,
                        int relu, int out_split, hipStream_t s, const float* resid, int ldr, int r_split, int x_f16) {
    if (rows <= 0) return 0;
    if (dim != 512 || (ld & 3) || (ldy & 3) || (resid && ((ldr & 3) || (r_split > 2 && r_split != 4))))
        return fail(-1, "layernorm: dim must be 512, ld a multiple of 4, residual fp32, split pairs or half rows");
    hipLaunchKernelGGL(layernorm512_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ld, y, ldy, rows, gamma, beta, relu, out_split,
                       resid, ldr, r_split, x_f16);
    VLSAT_LAUNCH_CHECK("layernorm512");
    return 0;
}

// rowscale[m] = scale / ||x[m, 0:512]||_2   (object heads: reference SGFN_MMG/model.py:327-330)
__global__ __launch_bounds__(256) void row_invnorm512_kernel(const float* __restrict__ x, int ld, int rows,
                                                             float scale, float* __restrict__ out, const float* __restrict__ x2, float* __restrict__ out2) {
    if (blockIdx.y != 0) { x = x2; out = out2; }                  // (the twin problem of a paired launch)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + (size_t)row * ld;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * lane);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 256 + 4 * lane);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += a[c] * a[c] + b[c] * b[c];
    s = wave_sum(s);
    if (lane == 0) out[row] = scale / sqrtf(s);
}
int launch_row_invnorm(const float* x, int ld, int rows, int dim, float scale, float* out, hipStream_t s, const float* x2, float* out2) {
    if (rows <= 0) return 0;
    if (dim != 512 || (ld & 3)) return fail(-1, "row_invnorm: dim must be 512 and ld a multiple of 4");
    if (!x2 != !out2) return fail(-1, "row_invnorm: a twin launch needs both the second input and the second output");
    hipLaunchKernelGGL(row_invnorm512_kernel, dim3((rows + 3) / 4, x2 ? 2 : 1), dim3(256), 0, s, x, ld, rows, scale, out, x2, out2);
    VLSAT_LAUNCH_CHECK("row_invnorm512");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Aggre_Index (reference network_util.py:64-73; aggr from MODEL.GCN_AGGR, flow target_to_source
// => reduce over the edges whose SOURCE is n).  CSR over sources, one block per node, one
// thread per channel: deterministic, no atomics; empty segment -> 0 (torch_scatter semantics).
// (blockIdx.y = 1: the twin problem of a paired launch -- gated2 / out2, same graph: round 6, one-scene plans)
__global__ __launch_bounds__(256) void aggregate_kernel(const float* __restrict__ gated, int n_ch,
                                                        const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ order, int aggr,
                                                        float* __restrict__ out, int ldo, int col0,
                                                        const float* __restrict__ gated2, float* __restrict__ out2) {
    if (blockIdx.y != 0) { gated = gated2; out = out2; }
    const int n = blockIdx.x;
    const int b = rowptr[n], e = rowptr[n + 1];
    for (int c = threadIdx.x; c < n_ch; c += 256) {          // (DIM_ATTEN = 512: two channels per thread)
        float acc = 0.f;
        if (e > b) {
            if (aggr == 0) {
                acc = -INFINITY;
                for (int k = b; k < e; ++k) acc = fmaxf(acc, gated[(size_t)order[k] * n_ch + c]);
            } else {
                for (int k = b; k < e; ++k) acc += gated[(size_t)order[k] * n_ch + c];
                if (aggr == 2) acc /= (float)(e - b);
            }
        }
        out[(size_t)n * ldo + col0 + c] = acc;
    }
}
// start values of the aggregation fused into the gate kernel (edge_gate_bf16.hip): -inf where a maximum will arrive, 0 for
// nodes without out-edges (torch_scatter's empty segment)
__global__ __launch_bounds__(256) void agg_init_kernel(const int32_t* __restrict__ rowptr, int n_nodes, int n_ch, float* __restrict__ agg, int ld, float* __restrict__ agg2) {
    if (blockIdx.y != 0) agg = agg2;
    const int i = blockIdx.x * 256 + threadIdx.x, n = i / (n_ch / 4), c4 = (i % (n_ch / 4)) * 4;
    if (n >= n_nodes) return;
    const float v = rowptr[n + 1] > rowptr[n] ? -INFINITY : 0.f;
    *reinterpret_cast<f32x4*>(agg + (size_t)n * ld + c4) = f32x4{v, v, v, v};
}
int launch_agg_init(const int32_t* rowptr, int n_nodes, int n_ch, float* agg, int ld_agg, hipStream_t s, float* agg2) {
    if (n_nodes <= 0) return 0;
    if ((n_ch & 3) || (ld_agg & 3)) return fail(-1, "agg_init: channel count and pitch must be multiples of 4");
    const long n4 = (long)n_nodes * (n_ch / 4);
    hipLaunchKernelGGL(agg_init_kernel, dim3((unsigned)((n4 + 255) / 256), agg2 ? 2 : 1), dim3(256), 0, s, rowptr, n_nodes, n_ch, agg, ld_agg, agg2);
    VLSAT_LAUNCH_CHECK("agg_init");
    return 0;
}

// p[0:n] = 0.  The library's own fill: hipMemsetAsync, driven from several host threads on several streams at once, was caught
// leaving foreign 8-byte patterns in the destination about once per 20 000 calls (tools/replica_race_probe.py,
// profiles/r04_probes/replica_race.txt); nothing on the forward path uses it any more.  V = 4: 16-byte stores (n a multiple
// of 4, p 16-byte aligned), V = 1 otherwise.
template <int V>
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, size_t nv) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        if constexpr (V == 4) reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        else p[i] = 0.f;
    }
}
int launch_zero_f32(float* p, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    const bool v4 = !((n & 3) || (reinterpret_cast<uintptr_t>(p) & 15));
    const size_t nv = v4 ? n / 4 : n;
    const dim3 grid((unsigned)std::min<size_t>((nv + 255) / 256, 2048));
    if (v4) hipLaunchKernelGGL(zero_f32_kernel<4>, grid, dim3(256), 0, s, p, nv);
    else hipLaunchKernelGGL(zero_f32_kernel<1>, grid, dim3(256), 0, s, p, nv);
    VLSAT_LAUNCH_CHECK("zero_f32");
    return 0;
}

// dst[r, 0:cols] = src[r, 0:cols] for r < rows (pitches in floats): the device-to-device copies of the forward path, for the
// same reason as zero_f32 (no runtime blit on a path that several host threads drive at once).  V = 4: 16-byte accesses
// (columns and pitches multiples of 4 floats, 16-byte aligned pointers), V = 1 otherwise.
template <int V>
__global__ __launch_bounds__(256) void copy_rows_kernel(float* __restrict__ dst, size_t dst_ld, const float* __restrict__ src, size_t src_ld,
                                                        int cv, size_t nv) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const size_t r = i / cv, c = (i % cv) * V;
        if constexpr (V == 4) *reinterpret_cast<f32x4*>(dst + r * dst_ld + c) = *reinterpret_cast<const f32x4*>(src + r * src_ld + c);
        else dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}
int launch_copy_rows(float* dst, size_t dst_ld, const float* src, size_t src_ld, int cols, size_t rows, hipStream_t s) {
    if (rows == 0 || cols <= 0) return 0;
    const bool v4 = !((cols & 3) || (dst_ld & 3) || (src_ld & 3) || ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15));
    const int cv = v4 ? cols / 4 : cols;
    const size_t nv = rows * (size_t)cv;
    const dim3 grid((unsigned)std::min<size_t>((nv + 255) / 256, 4096));
    if (v4) hipLaunchKernelGGL(copy_rows_kernel<4>, grid, dim3(256), 0, s, dst, dst_ld, src, src_ld, cv, nv);
    else hipLaunchKernelGGL(copy_rows_kernel<1>, grid, dim3(256), 0, s, dst, dst_ld, src, src_ld, cv, nv);
    VLSAT_LAUNCH_CHECK("copy_rows");
    return 0;
}

int launch_aggregate(const float* gated, int n_ch, const int32_t* rowptr, const int32_t* order, int n_nodes,
                     int aggr, float* out, int ldo, int col0, hipStream_t s, const float* gated2, float* out2) {
    if (n_nodes <= 0) return 0;
    if (!gated2 != !out2) return fail(-1, "aggregate: a twin launch needs both the second input and the second output");
    hipLaunchKernelGGL(aggregate_kernel, dim3(n_nodes, gated2 ? 2 : 1), dim3(256), 0, s, gated, n_ch, rowptr, order, aggr, out, ldo,
                       col0, gated2, out2);
    VLSAT_LAUNCH_CHECK("aggregate");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Distance-bias MLP of MMG.forward (reference network_MMG.py:190-203, self_attn_fc :165-173):
//   w = [c_b - c_a, ||c_b - c_a||] -> Linear(4,32) ReLU LN -> Linear(32,32) ReLU LN -> Linear(32,H)
// One thread per (scene, query a, key b); weights are read with wave-uniform indices (scalar loads).
// Output layout per scene: bias[bias_ptr[s] + (h*n + a)*n + b].
__device__ __forceinline__ void ln32(float (&t)[32], const float* g, const float* b) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) m += t[i];
    m *= (1.f / 32.f);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { t[i] -= m; v += t[i] * t[i]; }
    const float r = rsqrtf(v * (1.f / 32.f) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 32; ++i) t[i] = t[i] * r * g[i] + b[i];
}
__global__ __launch_bounds__(256) void dist_bias_kernel(const float* __restrict__ desc, int ld_desc,
                                                        const int32_t* __restrict__ scene_ptr,
                                                        const int64_t* __restrict__ bias_ptr, int n_heads,
                                                        DistBiasW w, float* __restrict__ bias) {
    const int s = blockIdx.y;
    const int n0 = scene_ptr[s], n = scene_ptr[s + 1] - n0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n) return;
    const int a = i / n, b = i % n;
    const float* ca = desc + (size_t)(n0 + a) * ld_desc;
    const float* cb = desc + (size_t)(n0 + b) * ld_desc;
    float x[4];
    x[0] = cb[0] - ca[0]; x[1] = cb[1] - ca[1]; x[2] = cb[2] - ca[2];
    x[3] = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    float t[32], u[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) {
        float acc = w.b0[o];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = fmaf(w.w0[o * 4 + k], x[k], acc);
        t[o] = fmaxf(acc, 0.f);
    }
    ln32(t, w.g2, w.be2);
#pragma unroll
    for (int o = 0; o < 32; ++o) {
        float acc = w.b3[o];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = fmaf(w.w3[o * 32 + k], t[k], acc);
        u[o] = fmaxf(acc, 0.f);
    }
    ln32(u, w.g5, w.be5);
    float* dst = bias + bias_ptr[s] + (size_t)a * n + b;
    for (int h = 0; h < n_heads; ++h) {
        float acc = w.b6[h];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = fmaf(w.w6[h * 32 + k], u[k], acc);
        dst[(size_t)h * n * n] = acc;
    }
}
int launch_dist_bias(const float* desc, int ld_desc, const int32_t* scene_ptr, const int64_t* bias_ptr,
                     int n_scenes, int max_n, int n_heads, DistBiasW w, float* bias, hipStream_t s) {
    if (n_scenes <= 0 || max_n <= 0) return 0;
    const int gx = (max_n * max_n + 255) / 256;
    hipLaunchKernelGGL(dist_bias_kernel, dim3(gx, n_scenes), dim3(256), 0, s, desc, ld_desc, scene_ptr, bias_ptr,
                       n_heads, w, bias);
    VLSAT_LAUNCH_CHECK("dist_bias");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Node attention: softmax(q.k*scale + bias) v per scene (block-diagonal mask of
// network_MMG.py:188-193 == per-scene loop) and head; d_k = 64.
// (reference transformer/attention.py:60-76 as called from network_MMG.py:217-218.)
// N per scene is small (40 at cfg 2, 200 at cfg 5): one query per lane, keys streamed through
// LDS in chunks of 64 (wave-uniform broadcast reads), online softmax, fp32 VALU.
template <int DK>
__global__ __launch_bounds__(64) void node_attn_kernel(const float* __restrict__ Q, int ldq,
                                                       const float* __restrict__ K, int ldk,
                                                       const float* __restrict__ V, int ldv,
                                                       float* __restrict__ O, int ldo,
                                                       const float* __restrict__ bias,
                                                       const int32_t* __restrict__ scene_ptr,
                                                       const int64_t* __restrict__ bias_ptr, float scale) {
    __shared__ __attribute__((aligned(16))) float sK[64 * DK];
    __shared__ __attribute__((aligned(16))) float sV[64 * DK];
    const int s = blockIdx.z, h = blockIdx.y, lane = threadIdx.x;
    const int n0 = scene_ptr[s], n = scene_ptr[s + 1] - n0;
    const int qa = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= n) return;
    const bool valid = qa < n;
    const int qrow = n0 + (valid ? qa : n - 1);
    float q[DK], acc[DK];
#pragma unroll
    for (int g = 0; g < DK / 4; ++g) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(Q + (size_t)qrow * ldq + h * DK + 4 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c) { q[4 * g + c] = x[c] * scale; acc[4 * g + c] = 0.f; }
    }
    float m_run = -INFINITY, l_run = 0.f;
    const float* brow = bias ? bias + bias_ptr[s] + ((size_t)h * n + (valid ? qa : n - 1)) * n : nullptr;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int kn = min(64, n - k0);
        __syncthreads();
        for (int i = lane; i < kn * (DK / 4); i += 64) {
            const int r = i / (DK / 4), c4 = (i % (DK / 4)) * 4;
            *reinterpret_cast<f32x4*>(sK + r * DK + c4) =
                *reinterpret_cast<const f32x4*>(K + (size_t)(n0 + k0 + r) * ldk + h * DK + c4);
            *reinterpret_cast<f32x4*>(sV + r * DK + c4) =
                *reinterpret_cast<const f32x4*>(V + (size_t)(n0 + k0 + r) * ldv + h * DK + c4);
        }
        __syncthreads();
        for (int j = 0; j < kn; ++j) {
            float sc = brow ? brow[k0 + j] : 0.f;
#pragma unroll
            for (int g = 0; g < DK / 4; ++g) {
                const f32x4 kk = *reinterpret_cast<const f32x4*>(sK + j * DK + 4 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) sc = fmaf(q[4 * g + c], kk[c], sc);
            }
            const float m_new = fmaxf(m_run, sc);
            const float alpha = __expf(m_run - m_new);
            const float p = __expf(sc - m_new);
            l_run = l_run * alpha + p;
            m_run = m_new;
#pragma unroll
            for (int g = 0; g < DK / 4; ++g) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(sV + j * DK + 4 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[4 * g + c] = fmaf(acc[4 * g + c], alpha, p * vv[c]);
            }
        }
    }
    if (valid) {
        const float inv = 1.f / l_run;
#pragma unroll
        for (int g = 0; g < DK / 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = acc[4 * g + c] * inv;
            *reinterpret_cast<f32x4*>(O + (size_t)qrow * ldo + h * DK + 4 * g) = o;
        }
    }
}
// The same attention for plans too small to fill the chip with one query per lane (a one-scene call: 8 heads x
// ceil(n / 64) waves on 256 CUs, each walking all n keys with 128 FMAs per key: 30 us at n = 40).  Sixteen lanes share a
// query: 4 split the head dim (partial dots meet through two quad shuffles), 4 take every fourth key each (own online
// softmax, merged at the end); 16 queries per 256-thread block.  K / V rows sit in LDS with a 16-byte pad so that the
// four key phases of a wave read disjoint banks.
template <int DK>
__global__ __launch_bounds__(256) void node_attn_split_kernel(const float* __restrict__ Q, int ldq,
                                                              const float* __restrict__ K, int ldk,
                                                              const float* __restrict__ V, int ldv,
                                                              float* __restrict__ O, int ldo,
                                                              const float* __restrict__ bias,
                                                              const int32_t* __restrict__ scene_ptr,
                                                              const int64_t* __restrict__ bias_ptr, float scale) {
    constexpr int DP = DK / 4, PITCH = DK + 4, CHUNK = 64;
    __shared__ __attribute__((aligned(16))) float sK[CHUNK * PITCH];
    __shared__ __attribute__((aligned(16))) float sV[CHUNK * PITCH];
    const int s = blockIdx.z, h = blockIdx.y, tid = threadIdx.x;
    const int n0 = scene_ptr[s], n = scene_ptr[s + 1] - n0;
    if (blockIdx.x * 16 >= n) return;
    const int part = tid & 3, ksi = (tid >> 2) & 3;
    const int qa = blockIdx.x * 16 + (tid >> 4);
    const bool valid = qa < n;
    const int qrow = n0 + (valid ? qa : n - 1);
    float q[DP], acc[DP];
#pragma unroll
    for (int g = 0; g < DP / 4; ++g) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(Q + (size_t)qrow * ldq + h * DK + part * DP + 4 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c) { q[4 * g + c] = x[c] * scale; acc[4 * g + c] = 0.f; }
    }
    float m_run = -INFINITY, l_run = 0.f;
    const float* brow = bias ? bias + bias_ptr[s] + ((size_t)h * n + (valid ? qa : n - 1)) * n : nullptr;
    for (int k0 = 0; k0 < n; k0 += CHUNK) {
        const int kn = min(CHUNK, n - k0);
        __syncthreads();
        for (int i = tid; i < kn * (DK / 4); i += 256) {
            const int r = i / (DK / 4), c4 = (i % (DK / 4)) * 4;
            *reinterpret_cast<f32x4*>(sK + r * PITCH + c4) = *reinterpret_cast<const f32x4*>(K + (size_t)(n0 + k0 + r) * ldk + h * DK + c4);
            *reinterpret_cast<f32x4*>(sV + r * PITCH + c4) = *reinterpret_cast<const f32x4*>(V + (size_t)(n0 + k0 + r) * ldv + h * DK + c4);
        }
        __syncthreads();
        for (int j = ksi; j < kn; j += 4) {
            float sc = 0.f;
#pragma unroll
            for (int g = 0; g < DP / 4; ++g) {
                const f32x4 kk = *reinterpret_cast<const f32x4*>(sK + j * PITCH + part * DP + 4 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) sc = fmaf(q[4 * g + c], kk[c], sc);
            }
            sc += __shfl_xor(sc, 1);
            sc += __shfl_xor(sc, 2);
            if (brow) sc += brow[k0 + j];
            const float m_new = fmaxf(m_run, sc);
            const float alpha = __expf(m_run - m_new);
            const float p = __expf(sc - m_new);
            l_run = l_run * alpha + p;
            m_run = m_new;
#pragma unroll
            for (int g = 0; g < DP / 4; ++g) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(sV + j * PITCH + part * DP + 4 * g);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[4 * g + c] = fmaf(acc[4 * g + c], alpha, p * vv[c]);
            }
        }
    }
    // merge the four key phases (lane bits 2 and 3); a phase that saw no key has m = -inf, l = 0
#pragma unroll
    for (int sh = 4; sh <= 8; sh <<= 1) {
        const float m_o = __shfl_xor(m_run, sh), l_o = __shfl_xor(l_run, sh);
        const float m_new = fmaxf(m_run, m_o);
        const float a = m_run == -INFINITY ? 0.f : __expf(m_run - m_new), b = m_o == -INFINITY ? 0.f : __expf(m_o - m_new);
        l_run = l_run * a + l_o * b;
#pragma unroll
        for (int d = 0; d < DP; ++d) acc[d] = acc[d] * a + __shfl_xor(acc[d], sh) * b;
        m_run = m_new;
    }
    if (valid && ksi == 0) {
        const float inv = 1.f / l_run;
#pragma unroll
        for (int g = 0; g < DP / 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = acc[4 * g + c] * inv;
            *reinterpret_cast<f32x4*>(O + (size_t)qrow * ldo + h * DK + part * DP + 4 * g) = o;
        }
    }
}
// d_k = 512 / NUM_HEADS: 64 (shipped), 32 or 128.  bias may be NULL (no additive term): the generic (VALU) path of the
// edge cross-attention for d_k != 64 runs through this kernel too, with the scenes' EDGE ranges as scene_ptr.
int launch_node_attn(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                     const float* bias, const int32_t* scene_ptr, const int64_t* bias_ptr, int n_scenes, int max_n,
                     int n_heads, int dk, float scale, hipStream_t s, int split_below) {
    if (n_scenes <= 0 || max_n <= 0) return 0;
    if ((ldq | ldk | ldv | ldo) & 3) return fail(-1, "node_attn: leading dims must be multiples of 4");
    const dim3 grid((max_n + 63) / 64, n_heads, n_scenes);
    // too few one-query-per-lane waves to occupy the chip: sixteen lanes per query instead
    if ((long)grid.x * n_heads * n_scenes < split_below) {
        const dim3 g16((max_n + 15) / 16, n_heads, n_scenes);
        switch (dk) {
            case 32: hipLaunchKernelGGL(node_attn_split_kernel<32>, g16, dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, bias, scene_ptr, bias_ptr, scale); break;
            case 64: hipLaunchKernelGGL(node_attn_split_kernel<64>, g16, dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, bias, scene_ptr, bias_ptr, scale); break;
            case 128: hipLaunchKernelGGL(node_attn_split_kernel<128>, g16, dim3(256), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, bias, scene_ptr, bias_ptr, scale); break;
            default: return fail(-1, "node_attn: head dim must be 32, 64 or 128");
        }
        VLSAT_LAUNCH_CHECK("node_attn");
        return 0;
    }
    switch (dk) {
        case 32: hipLaunchKernelGGL(node_attn_kernel<32>, grid, dim3(64), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, bias, scene_ptr, bias_ptr, scale); break;
        case 64: hipLaunchKernelGGL(node_attn_kernel<64>, grid, dim3(64), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, bias, scene_ptr, bias_ptr, scale); break;
        case 128: hipLaunchKernelGGL(node_attn_kernel<128>, grid, dim3(64), 0, s, Q, ldq, K, ldk, V, ldv, O, ldo, bias, scene_ptr, bias_ptr, scale); break;
        default: return fail(-1, "node_attn: head dim must be 32, 64 or 128");
    }
    VLSAT_LAUNCH_CHECK("node_attn");
    return 0;
}

}  // namespace vlsat

// ------------------------------------------------------------------------------------------
// One-time weight preparation of the split-bf16 GEMM path: w = hi + lo with both parts bf16
// (round-to-nearest-even), stored as raw 16-bit patterns.
namespace vlsat {
__global__ void split_bf16_kernel(const float* __restrict__ w, size_t n, uint16_t* __restrict__ hi,
                                  uint16_t* __restrict__ lo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    hi[i] = __builtin_bit_cast(uint16_t, h);
    lo[i] = __builtin_bit_cast(uint16_t, l);
}
// fp16 plane of a weight matrix (precision mode fp16_mixed): rne, clamped to the finite range
__global__ void to_f16_kernel(const float* __restrict__ w, size_t n, uint16_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = __builtin_bit_cast(uint16_t, (_Float16)fminf(fmaxf(w[i], -65504.f), 65504.f));
}
int launch_to_f16(const float* w, size_t n, uint16_t* out, hipStream_t s) {
    if (!n) return 0;
    hipLaunchKernelGGL(to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n, out);
    VLSAT_LAUNCH_CHECK("to_f16");
    return 0;
}
int launch_split_bf16(const float* w, size_t n, uint16_t* hi, uint16_t* lo, hipStream_t s) {
    if (!n) return 0;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, n, hi, lo);
    VLSAT_LAUNCH_CHECK("split_bf16");
    return 0;
}
}  // namespace vlsat
