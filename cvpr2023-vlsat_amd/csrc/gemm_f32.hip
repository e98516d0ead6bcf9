// fp32 MFMA GEMM with fused epilogues -- the engine behind every nn.Linear / Conv1d(k=1) of
// the path (reference call sites: network_MMG.py:40,93-98; attention.py:54-58,77;
// network_PointNet.py:329-339; SGFN_MMG/model.py:294,305-306,329-330; clip_adapter/model.py:27-29).
//
// C[M,N] = act(rowscale * (A[M,K] . W[N,K]^T) + bias + resid_scale*resid + g0[gi0] + g1[gi1])
//
// Block = 256 threads = 4 waves in a 2x2 grid; block tile BM x BN, wave tile (BM/2) x (BN/2)
// made of 32x32 MFMA tiles; K streamed in BK=32 slices through double-buffered LDS, one barrier
// per slice.  Default operand pipe (PipeF32Dma, gemm_core.h): the loads of slice t+1 are issued
// before the MFMAs of slice t as `buffer_load_dwordx4 ... lds` straight into the other buffer
// (XOR-swizzled unpadded rows, no VGPR round trip, no ds_write).  ReLU-on-A launches and operands
// beyond 32-bit offsets use the VGPR-staged pipe (PipeF32: global_load -> registers -> ds_write).
//
// PERSISTENT + pipelined across tiles: the grid is (at most) 2 blocks per CU and every block
// walks a list of output tiles.  The slice pipeline does not drain at a tile boundary: the
// first slice of the NEXT tile is prefetched under the last slice of the current one, the
// additive epilogue operands (residual / gathered rows) are loaded straight into the
// accumulators at the start of a tile, and the stores of a finished tile retire under the next
// tile's MFMAs.  With
// K = 512 (16 slices per tile) the per-tile load/store bubble was ~20 % of the tile time
// in the one-tile-per-block version (blocks co-resident on a CU run in lock-step, so their
// bubbles coincide instead of hiding each other).
//
// Tile order is XCD-aware: in round r the 64 blocks resident on XCD x (block id % 8 == x) own
// 64 CONSECUTIVE tiles (N fastest), i.e. whole M-panels, so an A panel is fetched from HBM
// once and re-read from that XCD's L2 by its N-tile neighbours.
//
// Tail: a launch covers only full rounds of the grid; the launcher hands the remaining
// M-panels to a second launch with a smaller tile so the last, mostly idle round of big tiles
// (up to 1 of 6 rounds at cfg 2) becomes a quarter-length round.
//
// Roofline: fp32 MFMA (157.3 TF).  Per 128x128x32 slice a block moves 32 KB from L2 for
// 1.05 MFLOP (~19 GB/s/CU at peak rate): MFMA-issue bound, LDS is 4 ds_read_b128 per 16 MFMAs.
#include <algorithm>
#include <cstdlib>
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

// ADD (compile time, so the 64 accumulator-init loads per lane are branch-free and batched):
//   bit 0 = residual, bit 1 = gathered rows g0, bit 2 = gathered rows g1.
// PREC: 0 = exact fp32 (PipeF32), 1 / 3 = bf16 / split-bf16 operands (PipeBF16, gemm_core.h).
// KSL: k-slices (of 32) moved per pipeline step.  2 for small single-round problems, which are bound
//      by the global-load round trip per step, not by MFMA issue: half the steps, twice the bytes in flight.
// TWIN: two problems of one shape in one launch, selected by blockIdx.y (see gemm_splitk.hip; single-round 64 x 64 launches only).
template <int BM, int BN, int ADD, int PREC, int KSL, bool TWIN = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs pa, GemmArgs pb, int n_tiles, int nbn) {
    const GemmArgs& p = (TWIN && blockIdx.y != 0) ? pb : pa;
    constexpr int TM = BM / 64, TN = BN / 64;
    using Pipe = typename PipeSel<BM, BN, PREC>::type;
    constexpr int SLICE = Pipe::STAGE_BYTES;
    constexpr int STAGE = SLICE * KSL;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g8 = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int KT = p.K / (BK * KSL);
    // (an experiment that started each row panel's k loop at a different slice, against L2 channel camping, measured no
    //  gain in any mode and cost the fp32 kernel 2 %: removed)
    auto kslice = [&](int, int kt) { return kt * (BK * KSL); };

    int round = 0;
    int v = xcd * g8 + slot;                 // tile of round r: (r*8 + xcd)*g8 + slot
    if (v >= n_tiles) return;
    const long long t_start = p.clock_probe ? clock64() : 0, w_start = p.clock_probe ? wall_clock64() : 0;
    int m0 = (v / nbn) * BM, n0 = (v % nbn) * BN;

    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    typename Pipe::Regs regs[KSL];
    const typename Pipe::Ctx ctx(p, tid);

#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) Pipe::load(ctx, p, m0, n0, kslice(m0, 0) + ks * BK, regs[ks], tid, smem + ks * SLICE);
#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) Pipe::store(smem + ks * SLICE, regs[ks], tid, p.relu_a);
    __syncthreads();

    // A-panel prefetch of the bf16 LDS-direct pipe (Pipe::PREFETCH): thread -> (line of the stage, part of the line)
    constexpr int PF_TPL = 256 / (BM * KSL) > 0 ? 256 / (BM * KSL) : 1;          // threads per 128-byte line
    const int pf_row = (tid / PF_TPL) % BM, pf_col = ((tid / PF_TPL) / BM) * BK + (tid % PF_TPL) * (32 / PF_TPL);
    const int pf_dist = Pipe::PREFETCH ? (p.prefetch < 0 ? 6 : p.prefetch) : 0;
    float pf_sink = 0.f;                     // destination of every prefetch touch: lives in one register for the whole kernel

    int buf = 0;
    while (true) {
        const int nv = ((round + 1) * 8 + xcd) * g8 + slot;
        const bool next_tile = nv < n_tiles;
        const int nm0 = (nv / nbn) * BM, nn0 = (nv % nbn) * BN;
        for (int kt = 0; kt < KT; ++kt) {
            char* cur = smem + buf * STAGE;
            char* nxt = smem + (buf ^ 1) * STAGE;
            const bool last = kt == KT - 1;
            const bool more = !last || next_tile;
            if (more) {
                const int lm0 = last ? nm0 : m0, ln0 = last ? nn0 : n0, lk = last ? kslice(nm0, 0) : kslice(m0, kt + 1);
#pragma unroll
                for (int ks = 0; ks < KSL; ++ks) Pipe::load(ctx, p, lm0, ln0, lk + ks * BK, regs[ks], tid, nxt + ks * SLICE);
            }
            bool touched = false;
            if (Pipe::PREFETCH && pf_dist > 0) {
                // slice kt + 1 + pf_dist of this block's (tile, k) sequence; it may belong to the next tile
                const int j = kt + 1 + pf_dist;
                const bool same = j < KT;
                if (same || next_tile) {
                    int jj = same ? j : j - KT;
                    jj = jj < KT ? jj : KT - 1;
                    int row = (same ? m0 : nm0) + pf_row;
                    row = row < p.M ? row : p.M - 1;
                    const float* src = p.A + (size_t)row * p.lda + kslice(same ? m0 : nm0, jj) + pf_col;
                    // hidden from hipcc's wait counting on purpose (it would drain it at the next barrier); the register is
                    // tied ("+v") so nothing else is ever allocated to it while a touch is in flight
                    asm volatile("global_load_dword %0, %1, off" : "+v"(pf_sink) : "v"(src) : "memory");
                    touched = true;
                }
            }
            // additive epilogue operands (residual / gathered rows) are loaded straight into the accumulators at the start
            // of a tile (C-in of the first MFMA): no extra registers, and the wait overlaps the co-resident block's MFMAs
            if (ADD != 0 && kt == 0) tile_init<TM, TN, ADD, (PREC == 0 || PREC == 4)>(p, m0, n0, wm, wn, lane, acc);
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) Pipe::mma(cur + ks * SLICE, wm, wn, acc, lane, p.relu_a);
            if constexpr (Pipe::PREFETCH) {
                // counted wait + raw barrier: __syncthreads() would make hipcc drain every outstanding load, the touch included
                if (touched) Pipe::store_keep1();
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                if (more) {
#pragma unroll
                    for (int ks = 0; ks < KSL; ++ks) Pipe::store(nxt + ks * SLICE, regs[ks], tid, p.relu_a);
                }
                __syncthreads();
            }
            if (last) {
                tile_epilogue<TM, TN, (PREC == 0 || PREC == 4)>(p, m0, n0, BM, BN, wm, wn, lane, acc);
                zero_acc<TM, TN>(acc);
            }
            buf ^= 1;
        }
        if (!next_tile) {
            if (p.clock_probe && tid == 0) {       // DVFS probe: shader cycles (s_memtime) vs 100 MHz wall clock per block
                long long* d = p.clock_probe + (size_t)blockIdx.x * 4;
                unsigned hw_id, xcc_id;                       // where the block ran: HW_ID (wave/simd/cu/sh/se), XCC_ID
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
                d[0] = clock64() - t_start; d[1] = wall_clock64() - w_start;
                d[2] = (round + 1) | ((long long)(hw_id & 0xffffu) << 16) | ((long long)(xcc_id & 0xfu) << 32); d[3] = 1;
            }
            break;
        }
        ++round;
        m0 = nm0;
        n0 = nn0;
    }
    if (Pipe::PREFETCH) {                    // no touch may outlive its register
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(pf_sink));
    }
}

double gemm_flops(const GemmArgs& a) { return 2.0 * a.M * (double)a.N * a.K; }

// resident 256-thread blocks the persistent grid may use (2 per CU): a constant of the device, cached per device id
static int slots() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cache[dev]) {
        int cus = 256;
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        cache[dev] = std::max(8, ((2 * cus) / 8) * 8);
    }
    return cache[dev];
}

template <int BM, int BN, int KSL = 1>
static int launch_t(const GemmArgs& a, int n_tiles, int grid, hipStream_t s) {
    const int nbn = (a.N + BN - 1) / BN;
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
#define VLSAT_GEMM_CASE(ADD, PREC) \
    case (PREC) * 8 + (ADD): hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, ADD, PREC, KSL>), dim3(grid), dim3(256), 0, s, a, a, n_tiles, nbn); break;
    // exact fp32 launches without ReLU-on-A take the LDS-direct staging pipe (internal precision code 4) when
    // the operands are addressable with 32-bit byte offsets and the additive mode is one the forward uses
    int prec = a.prec;
    const bool dma_ok = !a.no_dma && (add == 0 || add == 1 || add == 6) &&
                        ((size_t)a.M + 256) * a.lda * 4 < (1ull << 32) && ((size_t)a.N + 256) * a.ldw * 4 < (1ull << 32);
    if (prec == 0 && (a.a_split || a.r_split || a.c_split || a.c_scale != 1.f))
        return fail(-1, "gemm: operand formats and c_scale exist in the bf16 modes only");      // (the fp32 kernels fold them away)
    if (prec == 0 && dma_ok && !a.relu_a) prec = 4;
    if (a.a_split == 2 && !(prec == 1 && dma_ok)) return fail(-1, "gemm: half-row A needs the single-rounding bf16 precision and the LDS-direct pipe");
    if ((prec == 1 || prec == 3) && dma_ok) prec += a.a_split == 2 ? (a.half_f16 ? 14 : 12) : a.a_split ? 8 : 4;   // bf16 modes: A split on the fragment-read side, so ReLU-on-A is fine
    else if (a.a_split) return fail(-1, "gemm: split-pair A needs a bf16 precision and the LDS-direct pipe");
    switch (prec * 8 + add) {
        VLSAT_GEMM_CASE(0, 13) VLSAT_GEMM_CASE(1, 13) VLSAT_GEMM_CASE(6, 13)
        VLSAT_GEMM_CASE(0, 15) VLSAT_GEMM_CASE(6, 15)
        VLSAT_GEMM_CASE(0, 9) VLSAT_GEMM_CASE(1, 9) VLSAT_GEMM_CASE(6, 9)
        VLSAT_GEMM_CASE(0, 11) VLSAT_GEMM_CASE(1, 11) VLSAT_GEMM_CASE(6, 11)
        VLSAT_GEMM_CASE(0, 4) VLSAT_GEMM_CASE(1, 4) VLSAT_GEMM_CASE(6, 4)
        VLSAT_GEMM_CASE(0, 5) VLSAT_GEMM_CASE(1, 5) VLSAT_GEMM_CASE(6, 5)
        VLSAT_GEMM_CASE(0, 7) VLSAT_GEMM_CASE(1, 7) VLSAT_GEMM_CASE(6, 7)
        VLSAT_GEMM_CASE(0, 0) VLSAT_GEMM_CASE(1, 0) VLSAT_GEMM_CASE(2, 0) VLSAT_GEMM_CASE(3, 0)
        VLSAT_GEMM_CASE(4, 0) VLSAT_GEMM_CASE(5, 0) VLSAT_GEMM_CASE(6, 0) VLSAT_GEMM_CASE(7, 0)
        VLSAT_GEMM_CASE(0, 1) VLSAT_GEMM_CASE(1, 1) VLSAT_GEMM_CASE(6, 1)
        VLSAT_GEMM_CASE(0, 3) VLSAT_GEMM_CASE(1, 3) VLSAT_GEMM_CASE(6, 3)
        default: return fail(-1, "gemm: this precision / additive-operand combination is not built");
    }
#undef VLSAT_GEMM_CASE
    if (a.launches) ++*a.launches;         // a logical GEMM is a main launch plus (usually) a small-tile tail launch
    VLSAT_LAUNCH_CHECK("gemm_f32");
    return 0;
}

// rows [row0, M) of the problem as a sub-problem
static GemmArgs tail_of(const GemmArgs& a, int row0) {
    GemmArgs t = a;
    t.A += (size_t)row0 * a.lda;
    t.C += (size_t)row0 * a.ldc;
    t.M = a.M - row0;
    if (a.rowscale) t.rowscale += row0;
    if (a.resid) t.resid += (size_t)row0 * a.ldr;
    if (a.gi0) t.gi0 += row0;
    if (a.gi1) t.gi1 += row0;
    return t;
}

template <int BM, int BN>
static int run_tiled(const GemmArgs& a, hipStream_t s, int slot_mult2 = 2) {
    const int G = slots() / 2 * slot_mult2;            // resident block slots the grid may use (default: two per CU)
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    const long T = (long)nbm * nbn;
    if (T <= G) {                                   // one round: grid = tiles (rounded up to 8)
        const int grid = (int)((T + 7) / 8) * 8;
        // latency-bound: two k-slices per pipeline step (four per step measured no faster: tools/latency_probe.py, round 2)
        if (BM == 64 && BN == 64 && a.K % (2 * BK) == 0) return launch_t<64, 64, 2>(a, (int)T, grid, s);
        return launch_t<BM, BN>(a, (int)T, grid, s);
    }
    // full rounds with this tile; the remaining M-panels go to a smaller tile (see header)
    const long rounds = T / G;
    long main_panels = (rounds * G) / nbn;
    if (main_panels <= 0 || main_panels >= nbm || (BM == 64 && BN == 64)) return launch_t<BM, BN>(a, (int)T, G, s);
    GemmArgs m = a;
    m.M = (int)(main_panels * BM);
    int r = launch_t<BM, BN>(m, (int)(main_panels * nbn), G, s);
    if (r) return r;
    return launch_gemm(tail_of(a, (int)(main_panels * BM)), s);   // strictly fewer rows: terminates
}

static bool g_clock_probe_on();
// ---- two problems, one launch (one-scene plans, round 6) ----
static bool twin_shapes(const GemmArgs& a, const GemmArgs& b) {
    return a.M == b.M && a.N == b.N && a.K == b.K && a.lda == b.lda && a.ldw == b.ldw && a.ldc == b.ldc && a.ldr == b.ldr &&
           a.ldg0 == b.ldg0 && a.ldg1 == b.ldg1 && a.act == b.act && a.prec == b.prec && a.a_split == b.a_split && a.r_split == b.r_split &&
           a.c_split == b.c_split && a.c_scale == b.c_scale && a.resid_scale == b.resid_scale && !a.bias == !b.bias && !a.resid == !b.resid &&
           !a.g0 == !b.g0 && !a.g1 == !b.g1 && !a.rowscale == !b.rowscale && a.no_dma == b.no_dma && a.no_ring == b.no_ring &&
           a.no_p8 == b.no_p8 && a.k_rot == b.k_rot && a.c_f16_cols == b.c_f16_cols && a.g_f16 == b.g_f16 && a.half_f16 == b.half_f16 && !a.force_tile && !b.force_tile && !a.ablate && !b.ablate && a.prefetch == b.prefetch;
}
// single round of 64 x 64 tiles, two k-slices per step (what run_tiled<64, 64> launches for T <= G), grid.y = 2
static int launch_t_twin(const GemmArgs& a, const GemmArgs& b, int n_tiles, int grid, hipStream_t s) {
    const int nbn = (a.N + 63) / 64;
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
#define VLSAT_GEMM_CASE(ADD, PREC) \
    case (PREC) * 8 + (ADD): hipLaunchKernelGGL((gemm_f32_kernel<64, 64, ADD, PREC, 2, true>), dim3(grid, 2), dim3(256), 0, s, a, b, n_tiles, nbn); break;
    int prec = a.prec;
    const bool dma_ok = !a.no_dma && (add == 0 || add == 1 || add == 6) &&
                        ((size_t)a.M + 256) * a.lda * 4 < (1ull << 32) && ((size_t)a.N + 256) * a.ldw * 4 < (1ull << 32);
    if (prec == 0 && (a.a_split || a.r_split || a.c_split || a.c_scale != 1.f)) return 1;
    if (prec == 0 && dma_ok && !(a.relu_a || b.relu_a)) prec = 4;     // (ReLU-on-A of either problem: the VGPR-staged pipe for both -- same products)
    if (a.a_split == 2 && !(prec == 1 && dma_ok)) return 1;
    if ((prec == 1 || prec == 3) && dma_ok) prec += a.a_split == 2 ? (a.half_f16 ? 14 : 12) : a.a_split ? 8 : 4;
    else if (a.a_split) return 1;
    switch (prec * 8 + add) {
        VLSAT_GEMM_CASE(0, 13) VLSAT_GEMM_CASE(1, 13) VLSAT_GEMM_CASE(6, 13)
        VLSAT_GEMM_CASE(0, 15) VLSAT_GEMM_CASE(6, 15)
        VLSAT_GEMM_CASE(0, 9) VLSAT_GEMM_CASE(1, 9) VLSAT_GEMM_CASE(6, 9)
        VLSAT_GEMM_CASE(0, 11) VLSAT_GEMM_CASE(1, 11) VLSAT_GEMM_CASE(6, 11)
        VLSAT_GEMM_CASE(0, 4) VLSAT_GEMM_CASE(1, 4) VLSAT_GEMM_CASE(6, 4)
        VLSAT_GEMM_CASE(0, 5) VLSAT_GEMM_CASE(1, 5) VLSAT_GEMM_CASE(6, 5)
        VLSAT_GEMM_CASE(0, 7) VLSAT_GEMM_CASE(1, 7) VLSAT_GEMM_CASE(6, 7)
        VLSAT_GEMM_CASE(0, 0) VLSAT_GEMM_CASE(1, 0) VLSAT_GEMM_CASE(6, 0)
        default: return 1;
    }
#undef VLSAT_GEMM_CASE
    if (a.launches) ++*a.launches;
    VLSAT_LAUNCH_CHECK("gemm_f32 (pair)");
    return 0;
}

int launch_gemm_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t s) {
    // anything launch_gemm would refuse or treat specially stays with launch_gemm (the caller falls back to two launches)
    if (!a.A || !a.W || !a.C || !b.A || !b.W || !b.C || a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % BK || !twin_shapes(a, b)) return 1;
    if ((a.lda & 3) || (a.ldw & 3) || (a.prec != 0 && a.prec != 1 && a.prec != 3)) return 1;
    if (a.prec && (!a.Whi || !b.Whi || (a.prec == 3 && (!a.Wlo || !b.Wlo)) || (a.ldw & 7))) return 1;
    if (a.rowscale && (a.resid || a.g0 || a.g1)) return 1;
    for (const GemmArgs* q : {&a, &b})
        if ((reinterpret_cast<uintptr_t>(q->A) & 15) || (reinterpret_cast<uintptr_t>(q->W) & 15)) return 1;
    if (g_clock_probe_on()) return 1;
    const int G = slots();
    if (a.sk_ws && b.sk_ws) {
        const int r = launch_gemm_splitk(a, G, s, &b);
        if (r <= 0) return r;
    }
    // Mirror of launch_gemm's cascade for a problem this small: only the case that ends in ONE round of 64 x 64 tiles with two
    // k-slices per step is paired; everything else (8-phase partial rounds, ring kernel, wider tiles, several rounds) is not.
    const long T = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
    if (T > G || a.K % (2 * BK)) return 1;
    if (a.N % 256 == 0 && a.K % 128 == 0 && !a.no_dma && !a.no_ring && !a.no_p8 && !a.rowscale && (!a.c_f16_cols || a.c_f16_cols == a.N)) {           // 8-phase partial round?
        const bool p8_fmt = (a.prec == 1 && a.a_split == 2) || (a.prec == 3 && a.a_split == 1) || (a.prec == 0 && !a.a_split && !a.c_split && !a.r_split);
        const long panels = (a.M + 255) / 256, nbn = a.N / 256, part_min = a.p8_part_min > 0 ? a.p8_part_min : a.prec == 1 ? 32 : (G / 2 * 5) / 8;
        if (p8_fmt && panels * nbn >= part_min) return 1;
    }
    if ((a.prec == 1 || a.prec == 3) && !a.no_dma && !a.no_ring && a.N > 64 && !a.rowscale && (long)((a.M + 255) / 256) * ((a.N + 127) / 128) >= G / 2) return 1;   // ring kernel
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
    if (a.prec == 3 && !a.a_split && a.N >= 1024 && a.N <= 2048 && blocks(64, 128) <= G && blocks(64, 128) >= G / 2) return 1;
    if ((a.N > 64 && blocks(128, 128) >= G) || (a.N <= 64 && blocks(128, 64) >= G) || (a.N > 64 && blocks(64, 128) >= G)) return 1;
    return launch_t_twin(a, b, (int)T, (int)((T + 7) / 8) * 8, s);
}

static long long* g_clock_probe = nullptr;       // debug only (vlsat_debug_gemm_clock_probe): process-wide on purpose
void gemm_set_clock_probe(long long* buf) { g_clock_probe = buf; }
static bool g_clock_probe_on() { return g_clock_probe != nullptr; }

int launch_gemm(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    a.clock_probe = g_clock_probe;
    if (!a.A || !a.W || !a.C) return fail(-1, "gemm: null A/W/C");
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K <= 0 || a.K % BK) return fail(-1, "gemm: K must be a positive multiple of 32");
    if ((a.lda & 3) || (a.ldw & 3)) return fail(-1, "gemm: lda/ldw must be multiples of 4 floats");
    if (a.prec != 0 && a.prec != 1 && a.prec != 3) return fail(-1, "gemm: prec must be 0 (fp32), 1 (bf16) or 3 (bf16x3)");
    if (a.prec && (!a.Whi || (a.prec == 3 && !a.Wlo) || (a.ldw & 7))) return fail(-1, "gemm: bf16 path needs pre-split weights and ldw % 8 == 0");
    if (a.rowscale && (a.resid || a.g0 || a.g1))
        return fail(-1, "gemm: rowscale cannot be combined with resid/g0/g1 (additive operands are accumulator inits)");
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.W) & 15))
        return fail(-1, "gemm: A/W must be 16-byte aligned");
    if (a.half_f16 && (a.prec != 1 || a.a_split != 2 || a.c_split || a.r_split)) return fail(-1, "gemm: fp16 operands are half-row A launches of the single-rounding precision; their half-row output is c_f16_cols == N");
    if ((a.c_f16_cols || a.g_f16) && a.prec == 0) return fail(-1, "gemm: fp16 half-row columns / tables belong to the bf16 modes (the exact-fp32 kernels read and write fp32)");
    if (a.c_f16_cols && ((a.c_f16_cols != a.N && a.c_f16_cols % 256) || a.c_f16_cols > a.N || a.c_split)) return fail(-1, "gemm: c_f16_cols must be N or a multiple of 256 within N, of an fp32 output");
    if (a.g_f16 && (a.resid || !(a.g0 || a.g1) || a.N % 256 || ((a.ldg0 | a.ldg1) & 1) || ((reinterpret_cast<uintptr_t>(a.g0) | reinterpret_cast<uintptr_t>(a.g1)) & 7)))
        return fail(-1, "gemm: g_f16 needs gathered rows, no residual, N % 256 == 0 and 8-byte aligned tables");
    const int G = slots();
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
    if (a.sk_ws && !a.clock_probe) {                  // small launch: k range spread over otherwise idle CUs
        const int r = launch_gemm_splitk(a, G, s);
        if (r <= 0) return r;
    }
    // Large M; exact fp32, single-rounding bf16 with half-row operands or split-bf16 with split-pair operands: the full rounds of 256 x 256 tiles go to the 8-phase
    // kernel (gemm_bf16_p8.hip: one 8-wave block per CU), the remaining row panels to the kernels below
    if (((a.prec == 1 && a.a_split == 2) || (a.prec == 3 && a.a_split == 1) || (a.prec == 0 && !a.a_split && !a.c_split && !a.r_split)) && !a.no_dma && !a.no_ring &&
        !a.no_p8 && !a.rowscale && !a.clock_probe && (!a.c_f16_cols || (a.c_f16_cols == a.N && a.prec == 1 && !a.resid && (a.half_f16 || (!a.g0 && !a.g1 && !a.relu_a)))) && a.N % 256 == 0 && a.K % 128 == 0 &&
        ((size_t)a.M + 256) * a.lda * 4 < (1ull << 32) && ((size_t)a.M + 256) * a.ldc * 4 < (1ull << 32) &&
        ((size_t)a.N + 256) * a.ldw * 4 < (1ull << 32)) {
        const int G1 = G / 2;
        // (a last, partly filled panel rides along with a partial round: rows past M read as zeros through the buffer descriptors,
        //  their stores are dropped by them, additive operands clamp the row -- round 5: the 120-row remainder of the cfg 5 scene
        //  no longer is a launch of its own)
        const long nbn = a.N / 256, full = a.M / 256, panels = full + (a.M % 256 ? 1 : 0), rounds = full * nbn / G1;
        long main_panels = rounds * G1 / nbn;
        // less than one round left (the tail of a big launch, or a medium-sized one): a partial round costs a whole tile time
        // (one tile per CU), the 128 x 128 kernels ~0.7 (fp32) / ~0.5 (bf16) of it per full round of tiles -- from 5/8 of a
        // round on this kernel is the faster one
        // (single-rounding bf16: a tile is 17-30 us against 8 + 0.4-0.7 us per tile-equivalent on the small kernels -- from 32 tiles on
        //  the partial round wins; the cfg 5 scene's 7 032 remainder rows = 54 tiles took 27.7 us per launch on 64 x 64 tiles, as long
        //  as the full round in front of them: profiles/r05_cfg5_bf16_mixed_kernel_stats_serial.md)
        const long part_min = a.p8_part_min > 0 ? a.p8_part_min : a.prec == 1 ? 32 : (G1 * 5) / 8;          // tiles from which a partial round beats the small kernels
        // ... and from which the REMAINDER behind full rounds rides along as one more (balanced) round instead of a tail launch.  Round 6,
        // interleaved A/B at the bench batch (profiles/r06_probes/ab_p8_part_min_*.txt): single-rounding bf16 from 12 tiles on (the
        // 12 / 24 remainder tiles of every N = 512 / 1024 launch: bf16_mixed 10127-10139 -> 10518-10565 scenes/s, +4 % -- a fourth
        // round on 208 of the 256 CUs costs what the tail launch cost, but it is one dependent launch less per GEMM and leaves 48 CUs
        // to the other lanes); split-bf16 from 24 on (+1.2 %; with 12 only +0.5 %: its tiles are three times as long); exact fp32
        // keeps 5/8 of a round (24: -1.7 %, 12: -10 %: a tile is 131 us there)
        const long rem_min = a.p8_part_min > 0 ? a.p8_part_min : a.prec == 1 ? 12 : a.prec == 3 ? 24 : (G1 * 5) / 8;
        if (main_panels == 0 && panels * nbn >= part_min) main_panels = panels;
        // Full rounds followed by a remainder that would be a partial round of its own (the cfg 5 scene: 312 tiles = 1.2 rounds at
        // N = 512, 624 = 2.4 at N = 1024): ONE launch of rounds + 1 BALANCED rounds on T / (rounds + 1) blocks instead of a full and a
        // partial launch -- the same number of tile times, one launch skeleton less, and the CUs it leaves out are free for the other
        // lanes' kernels (round 5: kproj 41.7 -> 30.1 us, nn_edge.2 64.7 -> 48.6 at E = 39 800; cfg 5 step +3 %)
        if (rounds >= 1 && main_panels > 0 && main_panels < panels && (panels - main_panels) * nbn >= rem_min) {
            const long step = 8 * nbn, g2 = ((panels * nbn + rounds) / (rounds + 1) + step - 1) / step * step;
            if (g2 <= G1) {
                const int r = launch_gemm_p8(a, (int)(panels * nbn), (int)g2, s);
                if (r <= 0) return r;
            }
        }
        if (main_panels > 0) {
            GemmArgs m = a;
            m.M = (int)std::min<long>(main_panels * 256, a.M);
            const int r = launch_gemm_p8(m, (int)(main_panels * nbn), G1, s);
            if (r < 0) return r;
            if (r == 0) {
                if (m.M == a.M) return 0;
                return launch_gemm(tail_of(a, m.M), s);
            }
        }
    }
    // bf16 modes, large M: the full rounds go to the 3-stage ring kernel (gemm_bf16_ring.hip: one 8-wave block per CU,
    // 256 x 128 tiles, two slices in flight), the remaining row panels to the kernels below
    if ((a.prec == 1 || a.prec == 3) && !a.no_dma && !a.no_ring && a.N > 64 && !a.rowscale &&
        ((size_t)a.M + 256) * a.lda * 4 < (1ull << 32) && ((size_t)a.N + 256) * a.ldw * 4 < (1ull << 32)) {
        const int G1 = G / 2;
        // 128 x 256 tiles when N is a multiple of 256 and they still make full rounds (half the A bytes per flop), else 256 x 128
        for (int rbn = (a.ring_wide && a.N % 256 == 0) ? 256 : 128; rbn >= 128; rbn -= 128) {
            const int rbm = 32768 / rbn;
            const long nbm = (a.M + rbm - 1) / rbm, nbn = (a.N + rbn - 1) / rbn;
            const long rounds = nbm * nbn / G1;
            const long main_panels = rounds * G1 / nbn;
            if (main_panels <= 0) continue;
            GemmArgs m = a;
            m.M = (int)std::min<long>(main_panels * rbm, a.M);
            const int r = launch_gemm_ring(m, rbn, (int)(main_panels * nbn), G1, s);
            if (r < 0) return r;
            if (r == 0) {
                if (m.M == a.M) return 0;
                return launch_gemm(tail_of(a, m.M), s);
            }
            break;
        }
    }
    switch (a.force_tile) {                       // (experiment switch: the tile the sweep asks for)
        case 1: return run_tiled<128, 128>(a, s);
        case 2: return run_tiled<128, 64>(a, s);
        case 3: return run_tiled<64, 128>(a, s);
        case 4: return run_tiled<64, 64>(a, s);
        case 5: return run_tiled<64, 64>(a, s, 4);        // (experiment: four 64 x 64 blocks per CU)
        case 6: return run_tiled<64, 128>(a, s, 3);       // (experiment: three 64 x 128 blocks per CU)
        case 7: return run_tiled<64, 64>(a, s, 3);
        default: break;
    }
    // split-bf16 node-row launches with 1024..2048 output columns (self-attention QKV, cross-attention KV at the bench batch):
    // one round of 64 x 128 tiles beats two rounds of 64 x 64 by 6-8 us per launch (tools/gemm_tile_sweep.py, round 4:
    // 30.1 -> 24.2 us and 29.4 -> 22.0 us; every other node-row shape is best on what the rule below picks, fp32 within 2-4 us)
    if (a.prec == 3 && !a.a_split && a.N >= 1024 && a.N <= 2048 && blocks(64, 128) <= G && blocks(64, 128) >= G / 2)
        return run_tiled<64, 128>(a, s);
    // Largest tile that still gives every resident slot a tile; small problems (and the tails
    // of big ones) take smaller tiles so the launch covers as many CUs as the problem allows.
    if (a.N > 64 && blocks(128, 128) >= G) return run_tiled<128, 128>(a, s);
    if (a.N <= 64 && blocks(128, 64) >= G) return run_tiled<128, 64>(a, s);
    if (a.N > 64 && blocks(64, 128) >= G) return run_tiled<64, 128>(a, s);
    // exact fp32 on 64 x 64 tiles over more than one round of two blocks per CU (node rows of a batch: QKV 960 tiles, KV 640,
    // the node-side projection 2080): the kernel holds 80 VGPRs and 32 KB of LDS, so four blocks fit a CU and these latency-bound
    // launches take the wider grid -- KV 42.6 -> 31.0 us, QKV 46.5 -> 39.9, wnode 90.3 -> 78.9 (tools/gemm_tile_sweep.py, round 4)
    if (a.prec == 0 && blocks(64, 64) > G) return run_tiled<64, 64>(a, s, 4);
    return run_tiled<64, 64>(a, s);
}

}  // namespace vlsat
