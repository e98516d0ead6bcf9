// Eval ranking step that process_val runs on the forward's outputs:
//   evaluate_topk_object     reference src/utils/eva_utils_acc.py:27-39
//   evaluate_topk_predicate  :42-79
//   evaluate_triplet_topk    :137-213 (use_clip=True)
// The reference sorts per element on the CPU (for every edge it sorts the full 160x160x26 =
// 665 600-entry outer product: 74 ms/edge, ~116 s per cfg-2 scene).  A rank only needs COUNTS:
// walking a descending sort until `pred[gt] >= pred[idx] or index > topk` stops after
// min(#strictly-greater, topk) steps, and the position of the gt triple in the top-k list is
// #{conf > gt_conf} + 1.  The outer product is never materialised: 32 lanes per edge count
// the triples above each threshold along a staircase over the sorted class probabilities (below).
// All comparisons use the same fp32 products ((s*o)*r, no FMA) as the
// reference's two einsums, so counts are bit-exact functions of the fp32 inputs.
// Integer / HBM-latency work: no MFMA.
#include "common.h"
#include "kernels.h"

namespace vlsat {

// probs[n, :] = softmax(x[n, :])  (F.softmax in evaluate_triplet_topk :144); one wave per row
// out may alias x (every element is read and written by the same lane); log_out: log_softmax instead of softmax
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, int ld, int rows, int cols, float* out,
                                                           int log_out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + (size_t)row * ld;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += 64) m = fmaxf(m, p[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += expf(p[c] - m);
    s = wave_sum(s);
    if (log_out) {
        const float ls = logf(s);
        for (int c = lane; c < cols; c += 64) out[(size_t)row * cols + c] = (p[c] - m) - ls;
    } else {
        for (int c = lane; c < cols; c += 64) out[(size_t)row * cols + c] = expf(p[c] - m) / s;
    }
}

__global__ __launch_bounds__(256) void obj_rank_kernel(const float* __restrict__ pred, int ld, const int64_t* __restrict__ gt,
                                                       int n, int cols, int topk, int32_t* __restrict__ rank) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* p = pred + (size_t)row * ld;
    const float g = p[gt[row]];
    int c = 0;
    for (int k = lane; k < cols; k += 64) c += p[k] > g;
    c = (int)wave_sum((float)c);
    if (lane == 0) rank[row] = min(c, topk) + 1;
}

// sort ascending, subtract position (eva_utils_acc.py:73-77), n <= 32
__device__ inline void finish_edge(int* r, int n, int32_t* out, int stride_pad) {
    for (int i = 1; i < n; ++i) {
        int v = r[i], j = i - 1;
        while (j >= 0 && r[j] > v) { r[j + 1] = r[j]; --j; }
        r[j + 1] = v;
    }
    for (int i = 0; i < stride_pad; ++i) out[i] = i < n ? r[i] - i : 0;
}

// 32 lanes per edge (lane = relation class, R <= 32; coalesced reads of the edge's scores and labels), 8 edges per block.
// rank of gt class k = min(#{q : p[q] > p[k]}, topk) + 1; an edge without gt relation gets ONE rank from #{q : p[q] >= thr}
// (eva_utils_acc.py:55-61).  The edge's ranks are then sorted ascending and each loses its position (:73-77) -- by counting, the
// ranks being few: position of rank_k = #{gt j : rank_j < rank_k, or equal and j < k}.
__global__ __launch_bounds__(256) void rel_rank_kernel(const float* __restrict__ rel, const int64_t* __restrict__ gt_rel,
                                                       int n_edges, int R, int topk, float thr,
                                                       int32_t* __restrict__ out, int32_t* __restrict__ cnt) {
    __shared__ int s_sorted[8][32];
    const int tid = threadIdx.x, k = tid & 31, base = tid & 32;
    const int e = blockIdx.x * 8 + (tid >> 5);
    if (e >= n_edges) return;
    const bool in = k < R;
    const float v = in ? rel[(size_t)e * R + k] : -INFINITY;
    const bool is_gt = in && gt_rel[(size_t)e * R + k] == 1;
    const unsigned gts = (unsigned)(__ballot(is_gt) >> base);
    const unsigned ge = (unsigned)(__ballot(in && v >= thr) >> base);
    int c = 0;
    for (int q = 0; q < R; ++q) c += __shfl(v, base + q) > v;
    int rank = min(c, topk) + 1;
    int32_t* o = out + (size_t)e * R;
    if (gts == 0) {                                           // no gt relation: one rank
        const int g = __popc(ge);
        if (in) o[k] = k == 0 ? (g == R ? topk + 1 : g + 1) : 0;
        if (k == 0) cnt[e] = 1;
        return;
    }
    const int n = __popc(gts);
    int pos = 0;
    for (unsigned m = gts; m; m &= m - 1) {
        const int j = __ffs(m) - 1;
        const int rj = __shfl(rank, base + j);
        pos += rj < rank || (rj == rank && j < k);
    }
    // the sorted list through LDS: gt lane k owns slot `pos` (equal ranks hold distinct positions), lane i then writes slot i of the row
    if (is_gt) s_sorted[tid >> 5][pos] = rank - pos;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (one wave: its LDS writes are complete before its reads issue)
    if (in) o[k] = k < n ? s_sorted[tid >> 5][k] : 0;
    if (k == 0) cnt[e] = n;
}

// ---- triplet ranks by counting a STAIRCASE, not the outer product ---------------------------------------------------------------
// rank of a gt triple = min(#{(i,j,k): (s_i*o_j)*r_k > t}, topk) + 1 with s = probs[from], o = probs[to], t = (s_gt*o_gt)*r_gt.
// fp32 rounding is monotone, so with both probability rows sorted descending the pairs (i, j) that pass for a fixed relation k
// form a down-closed staircase: row i passes on a prefix of length len_k(i), and len_k is non-increasing in i.  A count below
// topk means every passing cell has (i+1)(j+1) <= count, so only the topk largest entries of each row matter: they are sorted
// ONCE PER NODE (N rows instead of E x C^2 products), and a lane per (edge, relation) walks its staircase -- a binary search on
// row 0, then rows whose length only shrinks -- until the count reaches topk or the staircase ends: <= 2 topk steps, typically
// a handful.  Every comparison is the reference's own fp32 product ((s*o)*r, eva_utils_acc.py:163-164), so the counts -- and the
// ranks -- are bit-exact functions of the inputs, equal to the brute-force count whatever the tie order of the sort.
// (Round 5's kernel walked the C^2 pairs of every edge with a pruning bound: 3.8 ms per branch on the 64-scene batch.)

// sorted[n, 0:K] = the K largest entries of probs[n, :] in descending order; one wave per node, rank by counting
__global__ __launch_bounds__(256) void sort_probs_kernel(const float* __restrict__ probs, int N, int C, int K, float* __restrict__ sorted) {
    extern __shared__ float s_row[];                               // [4][C]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = blockIdx.x * 4 + w;
    float* row = s_row + (size_t)w * C;
    if (n < N)
        for (int c = lane; c < C; c += 64) row[c] = probs[(size_t)n * C + c];
    __syncthreads();
    if (n >= N) return;
    for (int c = lane; c < C; c += 64) {
        const float v = row[c];
        int r = 0;
        for (int q = 0; q < C; ++q) {
            const float x = row[q];
            r += (x > v) || (x == v && q < c);
        }
        if (r < K) sorted[(size_t)n * K + r] = v;
    }
}

// 32 lanes per edge (lane = relation class, R <= 32), 8 edges per block
__global__ __launch_bounds__(256) void tri_rank_kernel(const float* __restrict__ probs, const float* __restrict__ sorted,
                                                       const float* __restrict__ rel, const int64_t* __restrict__ gt_cls,
                                                       const int64_t* __restrict__ gt_rel, const int64_t* __restrict__ edges, int E, int C,
                                                       int R, int K, int topk, float thr, int32_t* __restrict__ out) {
    extern __shared__ float s_so[];                                // [8][2][K] sorted rows + [8][32] ranks
    const int tid = threadIdx.x, grp = tid >> 5, k = tid & 31;
    const int e = blockIdx.x * 8 + grp;
    float* S = s_so + (size_t)grp * 2 * K;
    float* O = S + K;
    int* ranks = reinterpret_cast<int*>(s_so + (size_t)16 * K) + grp * 32;
    int a = 0, b = 0;
    if (e < E) {
        a = (int)edges[2 * (size_t)e]; b = (int)edges[2 * (size_t)e + 1];
        for (int c = k; c < K; c += 32) { S[c] = sorted[(size_t)a * K + c]; O[c] = sorted[(size_t)b * K + c]; }
    }
    __syncthreads();
    if (e >= E) return;
    const float r = k < R ? rel[(size_t)e * R + k] : 0.f;
    const bool is_gt = k < R && gt_rel[(size_t)e * R + k] == 1;
    const unsigned half = (unsigned)(__ballot(is_gt) >> (tid & 32));          // this edge's 32 lanes of the wave's 64-bit mask
    const float gs = __fmul_rn(probs[(size_t)a * C + gt_cls[a]], probs[(size_t)b * C + gt_cls[b]]);
    const int n = half ? __popc(half) : 1;
    const bool ge = half == 0;                                                // no gt relation: count conf >= thr (one rank)
    unsigned rest = half;
    const float s0 = S[0];
    for (int g = 0; g < n; ++g) {
        float t = thr;
        if (!ge) {
            const int kg = __ffs(rest) - 1;
            rest &= rest - 1;
            t = __fmul_rn(gs, __shfl(r, (tid & 32) + kg));
        }
        int cnt = 0;
        if (k < R) {
            auto pass = [&](float s, float o) {
                const float c = __fmul_rn(__fmul_rn(s, o), r);
                return ge ? c >= t : c > t;
            };
            int lo = 0, hi = K;                                               // row 0: first column that fails
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (pass(s0, O[mid])) lo = mid + 1; else hi = mid;
            }
            int len = lo;
            cnt = len;
            for (int i = 1; i < K && len > 0 && cnt < topk; ++i) {
                const float s = S[i];
                while (len > 0 && !pass(s, O[len - 1])) --len;
                cnt += len;
            }
            cnt = min(cnt, topk);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);          // (xor < 32: stays inside the edge's half)
        if (k == 0) ranks[g] = min(cnt, topk) + 1;
    }
    if (k == 0) finish_edge(ranks, n, out + (size_t)e * R, R);                // (lane 0 wrote them, lane 0 reads them)
}


// ---- the additive metrics vector of a batch (bench.py / dist.py: what the one all-reduce of the path carries) ----
// v = {n_scenes, N, E, sum obj3d, sum obj2d, sum rel3d, sum rel2d, #nodes whose 3D and 2D top-1 agree, #edges likewise},
// sums in fp64.  Two launches, no atomics: 256 blocks reduce rows block-strided to partials [256][6] in a fixed order, one
// block adds the partials in block order -- the result does not depend on timing.  (In PyTorch this is a dozen reductions,
// casts and nine scalar copies into the vector: 0.3 ms per step, 4 % of a bf16_mixed step.)
__device__ __forceinline__ void row_stats(const float* __restrict__ a, const float* __restrict__ b, int C, double& sa, double& sb, int& agree) {
    float ma = a[0], mb = b[0];
    int ia = 0, ib = 0;
    double xa = a[0], xb = b[0];
    for (int c = 1; c < C; ++c) {
        const float va = a[c], vb = b[c];
        xa += va; xb += vb;
        if (va > ma) { ma = va; ia = c; }        // first maximum wins, like torch.argmax
        if (vb > mb) { mb = vb; ib = c; }
    }
    sa += xa; sb += xb;
    agree += ia == ib;
}

__global__ __launch_bounds__(256) void checksum_partial_kernel(const float* __restrict__ o3, const float* __restrict__ o2, long N, int C,
                                                               const float* __restrict__ r3, const float* __restrict__ r2, long E, int R,
                                                               double* __restrict__ part) {
    // relation rows (R <= 32 floats, E of them): 256 rows at a time are one contiguous piece of each tensor -- staged through
    // LDS with coalesced loads, then one thread per row (a thread walking its own 104-byte row in HBM is latency-bound: 60 us)
    __shared__ float sa[256 * 32], sb[256 * 32];
    __shared__ double sh[6][256];
    double s[4] = {0, 0, 0, 0};
    int ag[2] = {0, 0};
    const int t = threadIdx.x;
    const long stride = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + t;
    {   // object rows (C = 160 floats): a wave per row, coalesced loads, fixed shuffle trees; lane 0 keeps the row's result
        const int lane = t & 63;
        const long wave0 = (long)blockIdx.x * 4 + (t >> 6), n_waves = (long)gridDim.x * 4;
        for (long n = wave0; n < N; n += n_waves) {
            const float* a = o3 + n * C;
            const float* b = o2 + n * C;
            double xa = 0, xb = 0;
            float ma = -INFINITY, mb = -INFINITY;
            int ia = 0x7fffffff, ib = 0x7fffffff;
            for (int c = lane; c < C; c += 64) {
                const float va = a[c], vb = b[c];
                xa += va; xb += vb;
                if (va > ma) { ma = va; ia = c; }      // (ascending c per lane: the first maximum of the lane's elements)
                if (vb > mb) { mb = vb; ib = c; }
            }
#pragma unroll
            for (int w = 32; w > 0; w >>= 1) {
                xa += __shfl_xor(xa, w); xb += __shfl_xor(xb, w);
                const float oa = __shfl_xor(ma, w), ob = __shfl_xor(mb, w);
                const int ja = __shfl_xor(ia, w), jb = __shfl_xor(ib, w);
                if (oa > ma || (oa == ma && ja < ia)) { ma = oa; ia = ja; }     // ties: the lowest class index, like torch.argmax
                if (ob > mb || (ob == mb && jb < ib)) { mb = ob; ib = jb; }
            }
            if (lane == 0) { s[0] += xa; s[1] += xb; ag[0] += ia == ib; }
        }
    }
    if (r3 && r2) {
        if (R <= 32) {
            for (long c0 = (long)blockIdx.x * 256; c0 < E; c0 += stride) {
                const int rows = (int)(E - c0 < 256 ? E - c0 : 256), nfl = rows * R;
                const float* pa = r3 + c0 * R;
                const float* pb = r2 + c0 * R;
                for (int i = t; i < nfl; i += 256) { sa[i] = pa[i]; sb[i] = pb[i]; }
                __syncthreads();
                if (t < rows) row_stats(sa + t * R, sb + t * R, R, s[2], s[3], ag[1]);
                __syncthreads();
            }
        } else {
            for (long e = t0; e < E; e += stride) row_stats(r3 + e * R, r2 + e * R, R, s[2], s[3], ag[1]);
        }
    }
    sh[0][t] = s[0]; sh[1][t] = s[1]; sh[2][t] = s[2]; sh[3][t] = s[3]; sh[4][t] = ag[0]; sh[5][t] = ag[1];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {            // fixed tree: thread t adds slot t + w
        if (t < w)
#pragma unroll
            for (int k = 0; k < 6; ++k) sh[k][t] += sh[k][t + w];
        __syncthreads();
    }
    if (t < 6) part[(size_t)blockIdx.x * 6 + t] = sh[t][0];
}

// n_blocks <= 256 partials per quantity, added by the same fixed tree
__global__ __launch_bounds__(256) void checksum_final_kernel(const double* __restrict__ part, int n_blocks, double n_scenes, double N, double E,
                                                             double* __restrict__ out) {
    __shared__ double sh[6][256];
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 6; ++k) sh[k][t] = t < n_blocks ? part[(size_t)t * 6 + k] : 0.0;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w)
#pragma unroll
            for (int k = 0; k < 6; ++k) sh[k][t] += sh[k][t + w];
        __syncthreads();
    }
    if (t < 6) out[3 + t] = sh[t][0];
    else if (t == 6) { out[0] = n_scenes; out[1] = N; out[2] = E; }
}

int launch_scene_checksums(const float* obj3d, const float* obj2d, long N, int C, const float* rel3d, const float* rel2d, long E, int R,
                           int n_scenes, double* out9, double* scratch, hipStream_t s) {
    if (C <= 0 || (E > 0 && R <= 0)) return fail(-1, "scene_checksums: class counts must be positive");
    constexpr int B = 256;
    hipLaunchKernelGGL(checksum_partial_kernel, dim3(B), dim3(256), 0, s, obj3d, obj2d, N, C, E > 0 ? rel3d : nullptr, E > 0 ? rel2d : nullptr, E, R, scratch);
    VLSAT_LAUNCH_CHECK("checksum_partial");
    hipLaunchKernelGGL(checksum_final_kernel, dim3(1), dim3(256), 0, s, scratch, B, (double)n_scenes, (double)N, (double)E, out9);
    VLSAT_LAUNCH_CHECK("checksum_final");
    return 0;
}

// ---- rank arrays -> the additive counts vector of evaluate.validation (what its one all-reduce carries) ----
// What MMGNet.validation derives from the concatenated rank lists of all scenes (reference src/model/model.py:214-242,
// 267-282), get_mean_recall (eva_utils_acc.py:224-237) and compute_mean_predicate (model.py:364-388) only needs COUNTS,
// additive over scenes (cvpr2023-vlsat_amd/evaluate.py: fields()):
//   [0] scenes | [1..R] cm_ge{k}: entries of the cls_matrix rows (sub_gt, sub_rank, obj_gt, obj_rank, predicate | -1) that are
//   >= k, clipped to R (the reference loops `range(int(cls_matrix.max()))` over the WHOLE matrix) | per branch (3D, 2D):
//   obj_n, obj_hit@{1,5,10}, rel_n, rel_hit@{1,3,5}, tri_n, tri_hit@{50,100}, then per predicate class c:
//   n, tri@{50,100}, rel@{1,3,5}.
// One thread per edge walks the edge's used rank slots (slot j = its j-th gt predicate in ascending class order, or the one
// "no relation" slot with predicate -1), one thread per node the object ranks; a block histogram in LDS, flushed with
// 64-bit integer atomics: the result does not depend on timing, so scenes may be accumulated from concurrent streams.
// The host path (evaluate.accumulate, numpy) stays as the reference-compatible one; this kernel keeps a one-scene-per-call
// evaluation loop free of host round trips.  Integer / latency work.
constexpr int CNT_MAX_R = 32;
__global__ __launch_bounds__(256) void eval_counts_kernel(const int32_t* __restrict__ obj_rank3, const int32_t* __restrict__ obj_rank2,
                                                          const int32_t* __restrict__ rel_rank3, const int32_t* __restrict__ rel_rank2,
                                                          const int32_t* __restrict__ tri_rank3, const int32_t* __restrict__ tri_rank2,
                                                          const int32_t* __restrict__ cnt, const int64_t* __restrict__ gt_cls,
                                                          const int64_t* __restrict__ gt_rel, const int64_t* __restrict__ edges,
                                                          int N, int E, int R, int n_scenes, unsigned long long* __restrict__ out) {
    extern __shared__ unsigned s_h[];                    // [1 + R + 2 * (11 + 6 R)] counters + [R + 1] histogram of cls_matrix entries
    const int per_br = 11 + 6 * R, n_out = 1 + R + 2 * per_br;
    unsigned* s_cm = s_h + n_out;
    for (int i = threadIdx.x; i < n_out + R + 1; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) s_h[0] = (unsigned)n_scenes;
    if (t < N) {
        const int32_t* ranks[2] = {obj_rank3, obj_rank2};
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            unsigned* b = s_h + 1 + R + br * per_br;
            const int r = ranks[br][t];
            atomicAdd(b + 0, 1u);
            if (r <= 1) atomicAdd(b + 1, 1u);
            if (r <= 5) atomicAdd(b + 2, 1u);
            if (r <= 10) atomicAdd(b + 3, 1u);
        }
    }
    if (t < E) {
        const int a = (int)edges[2 * (size_t)t], bn = (int)edges[2 * (size_t)t + 1];
        int head[4] = {(int)gt_cls[a], obj_rank3[a], (int)gt_cls[bn], obj_rank3[bn]};     // obj_topk = the 3D ranks for both branches
#pragma unroll
        for (int i = 0; i < 4; ++i) head[i] = head[i] < 0 ? 0 : head[i] > R ? R : head[i];
        const int n = cnt[t];
        const int64_t* g = gt_rel + (size_t)t * R;
        int k = -1;                                      // class of the current slot: the next set column of gt_rel[t], or -1
        for (int j = 0; j < n; ++j) {
            int pred = -1;
            for (int q = k + 1; q < R; ++q)
                if (g[q] == 1) { pred = q; break; }
            k = pred >= 0 ? pred : R;
#pragma unroll
            for (int i = 0; i < 4; ++i) atomicAdd(s_cm + head[i], 1u);
            atomicAdd(s_cm + (pred < 0 ? 0 : pred), 1u);                                   // (pred <= R - 1: never clipped from above)
            const int32_t* rr[2] = {rel_rank3, rel_rank2};
            const int32_t* tt[2] = {tri_rank3, tri_rank2};
#pragma unroll
            for (int br = 0; br < 2; ++br) {
                unsigned* b = s_h + 1 + R + br * per_br;
                const int r = rr[br][(size_t)t * R + j], tr = tt[br][(size_t)t * R + j];
                atomicAdd(b + 4, 1u);
                if (r <= 1) atomicAdd(b + 5, 1u);
                if (r <= 3) atomicAdd(b + 6, 1u);
                if (r <= 5) atomicAdd(b + 7, 1u);
                atomicAdd(b + 8, 1u);
                if (tr <= 50) atomicAdd(b + 9, 1u);
                if (tr <= 100) atomicAdd(b + 10, 1u);
                if (pred >= 0) {
                    unsigned* c = b + 11 + 6 * pred;
                    atomicAdd(c + 0, 1u);
                    if (tr <= 50) atomicAdd(c + 1, 1u);
                    if (tr <= 100) atomicAdd(c + 2, 1u);
                    if (r <= 1) atomicAdd(c + 3, 1u);
                    if (r <= 3) atomicAdd(c + 4, 1u);
                    if (r <= 5) atomicAdd(c + 5, 1u);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                              // cm_ge{k} = #entries >= k: suffix sums of the block's histogram
        unsigned run = 0;
        for (int v = R; v >= 1; --v) {
            run += s_cm[v];
            s_h[v] = run;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_out; i += blockDim.x)
        if (s_h[i]) atomicAdd(out + i, (unsigned long long)s_h[i]);
}

int launch_eval_counts(const int32_t* obj_rank3, const int32_t* obj_rank2, const int32_t* rel_rank3, const int32_t* rel_rank2,
                       const int32_t* tri_rank3, const int32_t* tri_rank2, const int32_t* cnt, const int64_t* gt_cls,
                       const int64_t* gt_rel, const int64_t* edges, int N, int E, int R, int n_scenes, unsigned long long* out,
                       hipStream_t s) {
    if (R <= 0 || R > CNT_MAX_R) return fail(-1, "eval_counts: at most 32 relation classes");
    const int n = N > E ? N : E;
    if (n <= 0 && n_scenes <= 0) return 0;
    const int n_out = 1 + R + 2 * (11 + 6 * R);
    const int grid = n > 0 ? (n + 255) / 256 : 1;
    hipLaunchKernelGGL(eval_counts_kernel, dim3(grid), dim3(256), (n_out + R + 1) * sizeof(unsigned), s, obj_rank3, obj_rank2, rel_rank3,
                       rel_rank2, tri_rank3, tri_rank2, cnt, gt_cls, gt_rel, edges, N, E, R, n_scenes, out);
    VLSAT_LAUNCH_CHECK("eval_counts");
    return 0;
}

int launch_softmax_rows(const float* x, int ld, int rows, int cols, float* out, int log_out, hipStream_t s) {
    if (rows <= 0) return 0;
    if (x == out && ld != cols) return fail(-1, "softmax_rows: in-place needs ld == cols");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ld, rows, cols, out, log_out);
    VLSAT_LAUNCH_CHECK("softmax_rows");
    return 0;
}

int launch_eval_ranks(const float* obj_logits, const float* obj_probs, const float* rel, const int64_t* gt_cls,
                      const int64_t* gt_rel, const int64_t* edges, int N, int E, int C, int R, int topk_obj,
                      int topk_rel, int topk_tri, float thr, int32_t* obj_rank, int32_t* rel_rank, int32_t* tri_rank,
                      int32_t* cnt, float* sorted_probs, hipStream_t s) {
    if (C > 1024 || R > 32) return fail(-1, "eval_ranks: at most 1024 object and 32 relation classes");
    if (topk_tri < 1) return fail(-1, "eval_ranks: topk_triplet must be positive");
    if (N > 0) {
        hipLaunchKernelGGL(obj_rank_kernel, dim3((N + 3) / 4), dim3(256), 0, s, obj_logits, C, gt_cls, N, C, topk_obj, obj_rank);
        VLSAT_LAUNCH_CHECK("obj_rank");
    }
    if (E > 0) {
        if (!sorted_probs) return fail(-1, "eval_ranks: null scratch (vlsat_eval_ranks_scratch_floats)");
        const int K = eval_ranks_sorted_k(C, topk_tri);
        if ((size_t)16 * K * sizeof(float) + 256 * sizeof(int) > 65536)      // (8 edges x 2 sorted rows of K floats per block; process_val's topk is 101)
            return fail(-1, "eval_ranks: min(n_obj_class, topk_triplet) must be at most 1008");
        hipLaunchKernelGGL(rel_rank_kernel, dim3((E + 7) / 8), dim3(256), 0, s, rel, gt_rel, E, R, topk_rel, thr, rel_rank, cnt);
        VLSAT_LAUNCH_CHECK("rel_rank");
        hipLaunchKernelGGL(sort_probs_kernel, dim3((N + 3) / 4), dim3(256), (size_t)4 * C * sizeof(float), s, obj_probs, N, C, K, sorted_probs);
        VLSAT_LAUNCH_CHECK("sort_probs");
        hipLaunchKernelGGL(tri_rank_kernel, dim3((E + 7) / 8), dim3(256), (size_t)16 * K * sizeof(float) + 256 * sizeof(int), s, obj_probs, sorted_probs, rel, gt_cls,
                           gt_rel, edges, E, C, R, K, topk_tri, thr, tri_rank);
        VLSAT_LAUNCH_CHECK("tri_rank");
    }
    return 0;
}

}  // namespace vlsat
