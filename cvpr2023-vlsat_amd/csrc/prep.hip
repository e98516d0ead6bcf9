// Per-object input preparation on the device -- what the reference's data
// loader does on the CPU for every object after sampling:
//   descriptor = gen_descriptor(sampled raw points)   reference src/utils/op_utils.py:47-64
//                [centroid(3), unbiased std(3), max-min(3), volume, max dim]
//   points     = sampled points - their mean           reference src/dataset/dataset_3dssg.py:189-191,292
//   layout     = [N,P,3] -> [N,3,P]                    reference src/model/model.py:79
// and the fully-connected edge list / batch ids of a batch of scenes
//   (dataset_3dssg.py:264-266, DataLoader.py:160-172).
// HBM-bound gather (12 B per sampled point in, 12 B out); one block per object, wave-shuffle
// reductions; statistics are two-pass (mean first, then centred sums), accumulated in fp64 like the reference's descriptor.
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

__device__ __forceinline__ double block_sum_f64(double v, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    return r;
}

// The reference builds the descriptor from the mesh vertices as trimesh hands them over -- float64 -- and stores it in a float32
// tensor (dataset_3dssg.py:273,290): the statistics here are accumulated in float64 and rounded once.  The object's points are
// converted to float32 first and centred with their float32 mean (:291-292); the mean used here is the float64 mean rounded to
// float32 (torch's float32 mean differs from it by at most an ulp of the mean).
__global__ __launch_bounds__(256) void prepare_objects_kernel(const float* __restrict__ scene, const int32_t* __restrict__ choice,
                                                              int P, float* __restrict__ obj_points,
                                                              float* __restrict__ desc) {
    __shared__ float red[4];
    __shared__ double red64[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int32_t* ch = choice + (size_t)n * P;
    double mean[3], dims[3], sd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double s = 0.0;
        float mx = -INFINITY, mn = INFINITY;
        for (int p = tid; p < P; p += 256) {
            const float v = scene[(size_t)ch[p] * 3 + c];
            s += (double)v; mx = fmaxf(mx, v); mn = fminf(mn, v);
        }
        mean[c] = block_sum_f64(s, red64) / (double)P;
        dims[c] = (double)block_reduce(mx, red, 1) - (double)block_reduce(mn, red, 2);
        const float mean32 = (float)mean[c];
        double q = 0.0;
        for (int p = tid; p < P; p += 256) {
            const float v = scene[(size_t)ch[p] * 3 + c];
            const double d = (double)v - mean[c];
            q += d * d;
            obj_points[((size_t)n * 3 + c) * P + p] = v - mean32;          // zero-meaned, [N,3,P]
        }
        sd[c] = sqrt(block_sum_f64(q, red64) / (double)(P - 1));           // unbiased (torch.std default)
    }
    if (tid == 0) {
        float* d = desc + (size_t)n * 11;
        d[0] = (float)mean[0]; d[1] = (float)mean[1]; d[2] = (float)mean[2];
        d[3] = (float)sd[0]; d[4] = (float)sd[1]; d[5] = (float)sd[2];
        d[6] = (float)dims[0]; d[7] = (float)dims[1]; d[8] = (float)dims[2];
        d[9] = (float)(dims[0] * dims[1] * dims[2]);
        d[10] = (float)fmax(dims[0], fmax(dims[1], dims[2]));
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

// ---- per-object point selection on the device (reference src/dataset/dataset_3dssg.py:279-289) -------------------------------------
//   obj_pointset = points[np.where(instances == instance_id)[0]];  choice = np.random.choice(len(obj_pointset), num_points, replace=True)
// Here: (1) the point indices of every requested instance as ONE segmented list in ascending point order (np.where's order) --
// a stable compaction: per 1024-point block and instance a count, an exclusive scan over blocks and instances, then every point
// writes its index at offset[instance] + prefix[block][instance] + its rank among the block's earlier points of that instance;
// (2) n_sample draws with replacement per instance from a COUNTER-BASED generator, so that a draw depends on (seed, object, draw
// number) only -- deterministic, any launch geometry, no state:
//   x = seed + 0x9E3779B97F4A7C15 * (obj * n_sample + draw + 1)   (mod 2^64)
//   z = splitmix64 finaliser of x:  z = (x ^ x >> 30) * 0xBF58476D1CE4E5B9;  z = (z ^ z >> 27) * 0x94D049BB133111EB;  z ^= z >> 31
//   k = ((z >> 32) * count) >> 32          (uniform on 0 .. count-1 up to count / 2^32)
//   choice[obj, draw] = list[offset[obj] + k]
// Not np.random's Mersenne stream (that is a property of the host library, not of the data path); oracle/prep_oracle.py restates
// this one bit for bit.  Integer work, HBM-bound (4 B per point read twice, 4 B written).
constexpr int SMP_CHUNK = 1024;      // points per block of the compaction (one wave walks them in 16 steps of 64)

__global__ void smp_map_kernel(int32_t* __restrict__ id_map, int map_size) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < map_size) id_map[i] = -1;
}
__global__ void smp_map_set_kernel(const int32_t* __restrict__ ids, int n_obj, int32_t* __restrict__ id_map, int map_size) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_obj && ids[i] >= 0 && ids[i] < map_size) id_map[ids[i]] = i;      // (duplicate ids: the last writer wins; callers pass a set)
}
// pass 0: block_cnt[block][slot] = points of instance `slot` in the block;  pass 1: list[...] = point indices, stable
template <int PASS>
__global__ __launch_bounds__(64) void smp_scan_kernel(const int32_t* __restrict__ inst, int64_t n_points, const int32_t* __restrict__ id_map,
                                                      int map_size, int n_obj, int32_t* __restrict__ block_cnt,
                                                      const int32_t* __restrict__ offset, int32_t* __restrict__ list) {
    extern __shared__ int32_t run[];                   // [n_obj]: counts so far in this block (pass 1: starting at the block's prefix)
    const int lane = threadIdx.x;
    int32_t* mine = block_cnt + (size_t)blockIdx.x * n_obj;
    for (int i = lane; i < n_obj; i += 64) run[i] = PASS ? mine[i] : 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SMP_CHUNK;
    for (int c = 0; c < SMP_CHUNK / 64; ++c) {
        const int64_t i = base + c * 64 + lane;
        int slot = -1;
        if (i < n_points) {
            const int id = inst[i];
            if (id >= 0 && id < map_size) slot = id_map[id];
        }
        // lanes of one instance are ranked together: leader = the first lane not yet served
        unsigned long long todo = __ballot(slot >= 0);
        while (todo) {
            const int lead = __builtin_ctzll(todo);
            const int ls = __shfl(slot, lead);
            const unsigned long long same = __ballot(slot == ls);
            if (slot == ls) {
                const int rank = __popcll(same & ((1ull << lane) - 1));
                if (PASS) list[(size_t)offset[ls] + run[ls] + rank] = (int32_t)i;
            }
            __syncthreads();                           // (one wave: orders the LDS read above against the update below)
            if (lane == lead) run[ls] += __popcll(same);
            __syncthreads();
            todo &= ~same;
        }
    }
    if (!PASS) for (int i = lane; i < n_obj; i += 64) mine[i] = run[i];
}
// per instance: exclusive scan of its block counts (in place) and its total; then the instances' offsets into the list
__global__ void smp_prefix_kernel(int32_t* __restrict__ block_cnt, int n_blocks, int n_obj, int32_t* __restrict__ counts,
                                  int32_t* __restrict__ offset) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_obj) {
        int32_t acc = 0;
        for (int b = 0; b < n_blocks; ++b) {
            const int32_t c = block_cnt[(size_t)b * n_obj + s];
            block_cnt[(size_t)b * n_obj + s] = acc;
            acc += c;
        }
        counts[s] = acc;
    }
}
__global__ void smp_offset_kernel(const int32_t* __restrict__ counts, int n_obj, int32_t* __restrict__ offset) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int32_t acc = 0;
        for (int s = 0; s < n_obj; ++s) { offset[s] = acc; acc += counts[s]; }
    }
}
__global__ void smp_draw_kernel(const int32_t* __restrict__ list, const int32_t* __restrict__ offset, const int32_t* __restrict__ counts,
                                int n_obj, int n_sample, unsigned long long seed, int32_t* __restrict__ choice) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_obj * n_sample) return;
    const int obj = (int)(t / n_sample);
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const unsigned cnt = (unsigned)counts[obj];
    const unsigned k = (unsigned)(((z >> 32) * (unsigned long long)cnt) >> 32);
    choice[t] = cnt ? list[(size_t)offset[obj] + k] : 0;       // (an instance without points: counts[obj] == 0 tells the caller)
}

size_t sample_objects_scratch_ints(int64_t n_points, int n_obj) {
    const size_t n_blocks = (size_t)((n_points + SMP_CHUNK - 1) / SMP_CHUNK);
    return n_blocks * (size_t)n_obj + (size_t)n_obj + (size_t)n_points;
}

int launch_sample_objects(const int32_t* instances, int64_t n_points, const int32_t* ids, int n_obj, int n_sample, unsigned long long seed,
                          int32_t* id_map, int map_size, int32_t* scratch, int32_t* choice, int32_t* counts, hipStream_t s) {
    if (n_obj <= 0 || n_sample <= 0) return 0;
    if (n_obj > 8192) return fail(-1, "sample_objects: at most 8192 instances per call (LDS table)");
    if (n_points <= 0 || n_points > 0x7fffffff || map_size <= 0) return fail(-1, "sample_objects: bad sizes");
    const int n_blocks = (int)((n_points + SMP_CHUNK - 1) / SMP_CHUNK);
    int32_t* block_cnt = scratch;
    int32_t* offset = scratch + (size_t)n_blocks * n_obj;
    int32_t* list = offset + n_obj;
    hipLaunchKernelGGL(smp_map_kernel, dim3((map_size + 255) / 256), dim3(256), 0, s, id_map, map_size);
    hipLaunchKernelGGL(smp_map_set_kernel, dim3((n_obj + 255) / 256), dim3(256), 0, s, ids, n_obj, id_map, map_size);
    hipLaunchKernelGGL(smp_scan_kernel<0>, dim3(n_blocks), dim3(64), n_obj * sizeof(int32_t), s, instances, n_points, id_map, map_size, n_obj,
                       block_cnt, offset, list);
    hipLaunchKernelGGL(smp_prefix_kernel, dim3((n_obj + 255) / 256), dim3(256), 0, s, block_cnt, n_blocks, n_obj, counts, offset);
    hipLaunchKernelGGL(smp_offset_kernel, dim3(1), dim3(64), 0, s, counts, n_obj, offset);
    hipLaunchKernelGGL(smp_scan_kernel<1>, dim3(n_blocks), dim3(64), n_obj * sizeof(int32_t), s, instances, n_points, id_map, map_size, n_obj,
                       block_cnt, offset, list);
    const size_t draws = (size_t)n_obj * n_sample;
    hipLaunchKernelGGL(smp_draw_kernel, dim3((unsigned)((draws + 255) / 256)), dim3(256), 0, s, list, offset, counts, n_obj, n_sample, seed, choice);
    VLSAT_LAUNCH_CHECK("sample_objects");
    return 0;
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
