// Split-K GEMM for the small launches of a one-scene forward (the reference's own call pattern: validation() runs
// batch_size = 1, src/model/model.py:185, so every nn.Linear sees 9..80 node rows or a few thousand edge rows).
//
// Why: a 64 x 64 output tile with K = 512 pulls 256 KB of operands through ONE CU, and a CU sustains 30-70 GB/s from
// L2 / Infinity Cache (tools/l2_fill_probe.hip) -- 15-19 us per launch whatever the problem size, with 16..300 of the 512
// resident slots in use (profiles/r02_single_scene_*: 52 such launches per forward).  Here the k range of a tile is cut
// into `ks` parts that run on different CUs (all parts of a tile on one XCD, so the partial sums meet in that XCD's L2);
// every part loads its whole k range with one round of LDS-direct loads, multiplies, writes its 64 x 64 partial sum to a
// workspace and bumps the tile's counter; the block that arrives last adds the parts IN PART ORDER (so the result does
// not depend on which block that is: same launch -> same bits) and runs the epilogue of the persistent kernel
// (gemm_f32.hip): accumulator init from residual / gathered rows, row scale, bias, activation, scale, output format.
// Soak: tools/splitk_soak.py (42 000 launches of six shapes, alone and next to a stream of large matmuls: every result
// bit-identical to the first); kernel tests: tests/test_hip_kernels.py::test_gemm_splitk_*.
#include <algorithm>
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

constexpr int SK_MAXS = 4;                   // k-slices (of 32) a block holds in LDS at once

// TWIN (round 6, one-scene plans): ONE launch for two problems of the same shape, flags and launch geometry -- the 3D / 2D twins of a
// GraphEdgeAttenNetwork block, of the relation encoders and of the heads -- selected by blockIdx.y, each with its own workspace
// and counters.  Every block runs exactly the code of the single launch on its own problem: the results are bit-identical to two
// launches; what goes away is one kernel's worth of start-up, drain and queue time per pair (a one-scene forward is ~114 such
// launches of ~10 us each, and a loop with several scenes in flight is bound by the SUM of their durations).
template <int PREC, bool TWIN = false>
__global__ __launch_bounds__(256, 2) void gemm_splitk_kernel(GemmArgs pa, GemmArgs pb, int n_tiles, int nbn, int ks, int slices, float* __restrict__ wsa,
                                                             unsigned* __restrict__ cnta, float* __restrict__ wsb, unsigned* __restrict__ cntb) {
    const bool second = TWIN && blockIdx.y != 0;
    const GemmArgs& p = second ? pb : pa;
    float* __restrict__ ws = second ? wsb : wsa;
    unsigned* __restrict__ counters = second ? cntb : cnta;
    using Pipe = typename PipeSel<64, 64, PREC>::type;
    constexpr int SLICE = Pipe::STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SK_MAXS * SLICE];
    __shared__ unsigned s_last;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tile = (j / ks) * 8 + xcd, part = j % ks;
    if (tile >= n_tiles) return;
    const int m0 = (tile / nbn) * 64, n0 = (tile % nbn) * 64;
    const int k_begin = part * slices * BK;
    const int n_sl = min(slices, (p.K - k_begin) / BK);

    f32x16 acc[1][1];
    zero_acc<1, 1>(acc);
    typename Pipe::Regs regs[SK_MAXS];
    const typename Pipe::Ctx ctx(p, tid);
    for (int c0 = 0; c0 < n_sl; c0 += SK_MAXS) {
        if (c0) __syncthreads();                                     // the previous chunk's fragments have been read
#pragma unroll
        for (int i = 0; i < SK_MAXS; ++i)
            if (c0 + i < n_sl) Pipe::load(ctx, p, m0, n0, k_begin + (c0 + i) * BK, regs[i], tid, smem + i * SLICE);
#pragma unroll
        for (int i = 0; i < SK_MAXS; ++i)
            if (c0 + i < n_sl) Pipe::store(smem + i * SLICE, regs[i], tid, p.relu_a);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SK_MAXS; ++i)
            if (c0 + i < n_sl) Pipe::mma(smem + i * SLICE, wm, wn, acc, lane, p.relu_a);
    }

    if (ks > 1) {
        // partial sums in accumulator order: element (wave, r, lane) -> one coalesced 256-byte store per wave and r
        float* mine = ws + ((size_t)tile * ks + part) * 4096 + wave * 1024 + lane;
        // Device-coherent accesses (sc1: written through / read past the XCD's L2) instead of fences: an agent-scope
        // release / acquire is a writeback + invalidate of the WHOLE L2 on this part (buffer_wbl2 / buffer_inv sc1), which
        // every one of the few hundred blocks would pay (measured: 55 us instead of 14 us per launch).
#pragma unroll
        for (int r = 0; r < 16; ++r) __hip_atomic_store(mine + r * 64, acc[0][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the stores are acknowledged before the counter moves
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(&counters[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ks - 1);
        __syncthreads();
        if (!s_last) return;
        if (tid == 0) __hip_atomic_store(&counters[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }

    // ---- accumulator init (the persistent kernel's additive operands), then the parts in order, then its epilogue ----
    f32x16 out[1][1];
    const int add = (p.resid ? 1 : 0) | (p.g0 ? 2 : 0) | (p.g1 ? 4 : 0);
    switch (add) {                                   // (the combinations the launcher of the persistent kernel knows too)
        case 1: tile_init<1, 1, 1>(p, m0, n0, wm, wn, lane, out); break;
        case 2: tile_init<1, 1, 2>(p, m0, n0, wm, wn, lane, out); break;
        case 3: tile_init<1, 1, 3>(p, m0, n0, wm, wn, lane, out); break;
        case 4: tile_init<1, 1, 4>(p, m0, n0, wm, wn, lane, out); break;
        case 5: tile_init<1, 1, 5>(p, m0, n0, wm, wn, lane, out); break;
        case 6: tile_init<1, 1, 6>(p, m0, n0, wm, wn, lane, out); break;
        case 7: tile_init<1, 1, 7>(p, m0, n0, wm, wn, lane, out); break;
        default: zero_acc<1, 1>(out);
    }
    if (ks > 1) {
        const float* parts = ws + (size_t)tile * ks * 4096 + wave * 1024 + lane;
        for (int s = 0; s < ks; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[0][0][r] += __hip_atomic_load(parts + (size_t)s * 4096 + r * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[0][0][r] += acc[0][0][r];
    }
    tile_epilogue<1, 1>(p, m0, n0, 64, 64, wm, wn, lane, out);
}

}  // namespace

// Decides whether the launch is one of the small ones this kernel is for and, if so, runs it.  Returns 0 = launched,
// 1 = not applicable (the caller falls through to the persistent kernel), < 0 = error.
// twin: a second problem of the same shape and flags (gemm_f32.hip launch_gemm_pair has checked that) for the same launch; every
// decision below is taken from `a` exactly as for a single launch, so each problem is computed as it would be alone.
int launch_gemm_splitk(const GemmArgs& a, int slots, hipStream_t s, const GemmArgs* twin) {
    if (!a.sk_ws || !a.sk_counters) return 1;
    if (twin && (!twin->sk_ws || !twin->sk_counters || twin->sk_ws == a.sk_ws || twin->sk_counters == a.sk_counters)) return 1;
    const long nbm = (a.M + 63) / 64, nbn = (a.N + 63) / 64, T = nbm * nbn;
    const int total = a.K / BK;                                      // k-slices
    if (total < 4 || T > slots / 2 || (a.sk_max_tiles > 0 && T > a.sk_max_tiles)) return 1;      // at least two parts of >= 2 slices, and room for them
    // as many parts as fill the resident slots once, each at least two slices (64 of K) long
    int ks = (int)std::min<long>(total / 2, std::max<long>(1, slots / T));
    ks = std::min(ks, 16);
    if (ks < 2) return 1;
    const int slices = (total + ks - 1) / ks;
    ks = (total + slices - 1) / slices;                              // no empty parts
    if ((size_t)T * ks * 4096 > a.sk_ws_floats || (size_t)T > a.sk_n_counters) return 1;
    if (twin && ((size_t)T * ks * 4096 > twin->sk_ws_floats || (size_t)T > twin->sk_n_counters)) return 1;
    int prec = a.prec;
    const bool relu_a = a.relu_a || (twin && twin->relu_a);          // (a pair takes the staging pipe that can apply ReLU to A if either needs it: same products)
    const bool dma_ok = !a.no_dma && ((size_t)a.M + 256) * a.lda * 4 < (1ull << 32) && ((size_t)a.N + 256) * a.ldw * 4 < (1ull << 32);
    if (prec == 0 && dma_ok && !relu_a) prec = 4;
    if (a.a_split == 2 && !(prec == 1 && dma_ok)) return 1;
    if ((prec == 1 || prec == 3) && dma_ok) prec += a.a_split == 2 ? (a.half_f16 ? 14 : 12) : a.a_split ? 8 : 4;
    else if (a.a_split) return 1;
    const int grid = (int)((T + 7) / 8) * 8 * ks;
    const GemmArgs& b = twin ? *twin : a;
#define VLSAT_SK_CASE(PREC) \
    case PREC: \
        if (twin) hipLaunchKernelGGL((gemm_splitk_kernel<PREC, true>), dim3(grid, 2), dim3(256), 0, s, a, b, (int)T, (int)nbn, ks, slices, a.sk_ws, a.sk_counters, b.sk_ws, b.sk_counters); \
        else hipLaunchKernelGGL((gemm_splitk_kernel<PREC, false>), dim3(grid), dim3(256), 0, s, a, a, (int)T, (int)nbn, ks, slices, a.sk_ws, a.sk_counters, a.sk_ws, a.sk_counters); \
        break;
    switch (prec) {
        VLSAT_SK_CASE(0) VLSAT_SK_CASE(1) VLSAT_SK_CASE(3) VLSAT_SK_CASE(4) VLSAT_SK_CASE(5) VLSAT_SK_CASE(7)
        VLSAT_SK_CASE(9) VLSAT_SK_CASE(11) VLSAT_SK_CASE(13) VLSAT_SK_CASE(15)
        default: return 1;
    }
#undef VLSAT_SK_CASE
    if (a.launches) ++*a.launches;
    VLSAT_LAUNCH_CHECK("gemm_splitk");
    return 0;
}

}  // namespace vlsat
