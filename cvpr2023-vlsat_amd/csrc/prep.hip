// Per-object input preparation on the device -- what the reference's data
// loader does on the CPU for every object after sampling:
//   descriptor = gen_descriptor(sampled raw points)   reference src/utils/op_utils.py:47-64
//                [centroid(3), unbiased std(3), max-min(3), volume, max dim]
//   points     = sampled points - their mean           reference src/dataset/dataset_3dssg.py:189-191,292
//   layout     = [N,P,3] -> [N,3,P]                    reference src/model/model.py:79
// and the fully-connected edge list / batch ids of a batch of scenes
//   (dataset_3dssg.py:264-266, DataLoader.py:160-172).
// HBM-bound gather (12 B per sampled point in, 12 B out); one block per object, wave-shuffle
// reductions; statistics are two-pass (mean first, then centred sums) in fp32.
#include "common.h"
#include "kernels.h"

namespace vlsat {

__device__ __forceinline__ float block_reduce(float v, float* red, int op) {   // op 0 sum, 1 max, 2 min
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o);
        v = op == 0 ? v + t : (op == 1 ? fmaxf(v, t) : fminf(v, t));
    }
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = op == 0 ? r + red[i] : (op == 1 ? fmaxf(r, red[i]) : fminf(r, red[i]));
    return r;
}

__global__ __launch_bounds__(256) void prepare_objects_kernel(const float* __restrict__ scene, const int32_t* __restrict__ choice,
                                                              int P, float* __restrict__ obj_points,
                                                              float* __restrict__ desc) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int32_t* ch = choice + (size_t)n * P;
    float mean[3], dims[3], sd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f, mx = -INFINITY, mn = INFINITY;
        for (int p = tid; p < P; p += 256) {
            const float v = scene[(size_t)ch[p] * 3 + c];
            s += v; mx = fmaxf(mx, v); mn = fminf(mn, v);
        }
        mean[c] = block_reduce(s, red, 0) / (float)P;
        dims[c] = block_reduce(mx, red, 1) - block_reduce(mn, red, 2);
        float q = 0.f;
        for (int p = tid; p < P; p += 256) {
            const float d = scene[(size_t)ch[p] * 3 + c] - mean[c];
            q += d * d;
            obj_points[((size_t)n * 3 + c) * P + p] = d;                   // zero-meaned, [N,3,P]
        }
        sd[c] = sqrtf(block_reduce(q, red, 0) / (float)(P - 1));          // unbiased (torch.std default)
    }
    if (tid == 0) {
        float* d = desc + (size_t)n * 11;
        d[0] = mean[0]; d[1] = mean[1]; d[2] = mean[2];
        d[3] = sd[0]; d[4] = sd[1]; d[5] = sd[2];
        d[6] = dims[0]; d[7] = dims[1]; d[8] = dims[2];
        d[9] = dims[0] * dims[1] * dims[2];
        d[10] = fmaxf(dims[0], fmaxf(dims[1], dims[2]));
    }
}

// edges [2,E] int64 (row 0 = from, row 1 = to, source-major, offsets applied) and batch_ids [N]
__global__ void fc_edges_kernel(const int32_t* __restrict__ node_ptr, const int64_t* __restrict__ edge_ptr, int n_scenes,
                                int64_t n_edges, int64_t* __restrict__ edges, int64_t* __restrict__ batch_ids) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < node_ptr[n_scenes]) {                                         // first N threads also write batch ids
        int lo = 0, hi = n_scenes - 1;
        while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (node_ptr[m] <= e) lo = m; else hi = m - 1; }
        batch_ids[e] = lo;
    }
    if (e >= n_edges) return;
    int lo = 0, hi = n_scenes - 1;
    while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (edge_ptr[m] <= e) lo = m; else hi = m - 1; }
    const int n = node_ptr[lo + 1] - node_ptr[lo];
    const int64_t l = e - edge_ptr[lo];
    const int i = (int)(l / (n - 1)), jj = (int)(l % (n - 1));
    edges[e] = node_ptr[lo] + i;
    edges[n_edges + e] = node_ptr[lo] + jj + (jj >= i);
}

// mismatches += #{e : (edges[e], edges[E + e]) != (src[e], dst[e])} + #{i : batch_ids starts a new run at i  XOR  i is a scene start}
__global__ void check_graph_kernel(const int64_t* __restrict__ edges, int64_t n_edges, const int32_t* __restrict__ src,
                                   const int32_t* __restrict__ dst, const int64_t* __restrict__ batch_ids, int64_t n_nodes,
                                   const int32_t* __restrict__ scene_ptr, int n_scenes, int32_t* __restrict__ mismatches) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    if (i < n_edges) bad += edges[i] != (int64_t)src[i] || edges[n_edges + i] != (int64_t)dst[i];
    if (batch_ids && i < n_nodes) {
        int lo = 0, hi = n_scenes - 1;
        while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (scene_ptr[m] <= i) lo = m; else hi = m - 1; }
        const bool start = scene_ptr[lo] == i;
        bad += i > 0 && ((batch_ids[i] != batch_ids[i - 1]) != start);
    }
    const unsigned long long any = __ballot(bad != 0);
    if (any && (threadIdx.x & 63) == 0) atomicAdd(mismatches, (int)__popcll(any));
}

int launch_check_graph(const int64_t* edges, int64_t n_edges, const int32_t* src, const int32_t* dst, const int64_t* batch_ids,
                       int64_t n_nodes, const int32_t* scene_ptr, int n_scenes, int32_t* mismatches, hipStream_t s) {
    const int64_t work = n_edges > n_nodes ? n_edges : n_nodes;
    if (work <= 0) return 0;
    hipLaunchKernelGGL(check_graph_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, edges, n_edges, src, dst,
                       batch_ids, n_nodes, scene_ptr, n_scenes, mismatches);
    VLSAT_LAUNCH_CHECK("check_graph");
    return 0;
}

int launch_prepare_objects(const float* scene, const int32_t* choice, int N, int P, float* obj_points, float* desc,
                           hipStream_t s) {
    if (N <= 0) return 0;
    if (P < 2) return fail(-1, "prepare_objects: need at least 2 points per object (unbiased std)");
    hipLaunchKernelGGL(prepare_objects_kernel, dim3(N), dim3(256), 0, s, scene, choice, P, obj_points, desc);
    VLSAT_LAUNCH_CHECK("prepare_objects");
    return 0;
}

int launch_fc_edges(const int32_t* node_ptr, const int64_t* edge_ptr, int n_scenes, int64_t n_nodes, int64_t n_edges,
                    int64_t* edges, int64_t* batch_ids, hipStream_t s) {
    const int64_t work = n_edges > n_nodes ? n_edges : n_nodes;
    if (work <= 0) return 0;
    hipLaunchKernelGGL(fc_edges_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, node_ptr, edge_ptr, n_scenes,
                       n_edges, edges, batch_ids);
    VLSAT_LAUNCH_CHECK("fc_edges");
    return 0;
}

}  // namespace vlsat
