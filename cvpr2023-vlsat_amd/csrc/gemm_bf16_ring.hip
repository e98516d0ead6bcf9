// bf16 / split-bf16 GEMM for the large edge-row launches of the bf16 modes (BASELINE configs[2]): same contract as
// gemm_f32.hip,
//     C[M,N] = act((A[M,K] . W[N,K]^T) + bias + resid_scale*resid + g0[gi0] + g1[gi1]) * c_scale,
// but built around what bounds a bf16 GEMM on this chip when its operands are 4 bytes per element: not the matrix
// pipe (a 128x128x32 slice is 768 cycles of v_mfma_f32_32x32x16_bf16 per wave in split-bf16, 256 in plain bf16) and not
// L2 bandwidth, but the ~1.2-1.5 us an LDS-direct load takes from issue to landing.  The two-stage kernel of
// gemm_f32.hip keeps one slice per block in flight and plateaus at ~29 GB/s per CU whatever the staging method
// (tools/gemm_bench.py: VGPR-staged 215 TF, LDS-direct + split on read 220 TF, split-pair operands 215 TF, the same
// with the operands L2-resident).  Little's law: the MFMA-bound rate needs ~100 KB in flight per CU.
//
// Structure: ONE persistent block of 8 waves per CU (4 x 2 waves, wave tile 64 x 64, block tile 256 x 128), BK = 32,
// a THREE-stage LDS ring (3 x 48 KB) filled by `buffer_load_dwordx4 ... lds`; two slices (96 KB) are always in flight:
//     step g:  s_waitcnt vmcnt(loads of ONE slice)   -> this wave's part of slice g has landed, g+1 may still fly
//              s_barrier                             -> everyone's part landed; everyone is done reading slice g-1
//              issue slice g+2 into the ring slot slice g-1 occupied
//              MFMAs of slice g
// i.e. one barrier per slice, counted waits, never a drain (cdna_hip_programming.md T3/T4).  The slice sequence runs on
// across tile boundaries (the issue cursor is two slices ahead of the compute cursor, in the next tile if need be).
// Operands: A fp32 or split-pair words (common.h pack_split), 128-byte rows XOR-swizzled by (row>>1)&7; pre-split
// weight planes, 64-byte rows swizzled by (row>>2)&3 -- as PipeSplitDma (gemm_core.h), whose fragment code this reuses.
// Tile order is XCD-aware exactly like gemm_f32.hip; the launcher (launch_gemm) hands this kernel the full rounds of a
// launch and the remaining row panels to the 128 x 128 kernel.
// Roofline: bf16 MFMA, 2.5 PF / TERMS.
#include <type_traits>

#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {


template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Block tile RBM x RBN = 256 x 128 (waves 4 x 2; the default) or 128 x 256 (waves 2 x 4; GemmArgs::ring_wide); wave tile
// 64 x 64 either way.  The second shape halves the bytes of A (read once, from HBM / Infinity Cache) per flop and doubles
// those of the weights (re-read from L2); it measured the same time on every launch of the forward -- the fill path does
// not care where the bytes come from -- and is kept as an experiment switch only.
// RBK: k per slice, 32 or -- half-row A with one weight plane only -- 64: the single-rounding modes issue just 8 MFMAs per
// wave between two barriers at 32 (ablation: 0.58 us per step of pure synchronisation against 0.2 us of MFMAs), and their
// 24 KB stages leave room for slices twice as long.  With 64 both operands have 128-byte rows in LDS, laid out like the
// fp32 A rows (XOR swizzle (row >> 1) & 7 over eight 16-byte chunks).
// DB (with RBK = 64, K a multiple of 128): the fragments of a slice are read from LDS one pipeline step EARLY, into a second
// register set, while the matrix pipe works on the previous slice's set -- LDS reads and MFMAs of a wave overlap across
// the barrier instead of alternating (ablation: either phase alone takes ~45 % of the full kernel's time above the store
// floor, together 60 %).  The loop is unrolled by two so that the two sets are compile-time registers.
template <int TERMS, int AFMT, int ADD, int RBN, int RBK = 32, bool DB = false>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(GemmArgs p, int n_tiles, int nbn) {
    static_assert(RBK == 32 || (RBK == 64 && AFMT >= 2 && TERMS == 1), "64-wide slices: half-row A, one plane");
    static_assert(!DB || RBK == 64, "double-buffered fragments: built for the 64-wide slices");
    // (the split-bf16 variants have no registers left for a second fragment set: 76-197 spilled registers when tried)
    constexpr int BK = RBK;                                  // (shadows the library-wide slice length inside this kernel)
    constexpr bool LR = RBK == 64;                           // long rows: 128 bytes per operand row and slice
    constexpr int RBM = 32768 / RBN;
    using Frag = PipeSplitDma<128, 128, TERMS, AFMT>;        // fragment-side helpers only (split8 / frag_half)
    constexpr bool AH = AFMT >= 2;                           // A as half rows (bf16; 3: fp16 operands, GemmArgs::half_f16): 64-byte slices like the weight planes
    constexpr int PL = TERMS == 1 ? 1 : 2;
    constexpr int TM = 2, TN = 2, WR = LR ? RBN / 64 : RBN / 128;      // (WR: instruction rounds of a weight-plane slice)
    constexpr int AR = (AH && !LR) ? RBM / 128 : RBM / 64;      // instruction rounds of an A slice (128 | 64 rows each)
    constexpr int A_BYTES = AH ? RBM * BK * 2 : RBM * BK * 4, W_PLANE = RBN * BK * 2;
    constexpr int STAGE = A_BYTES + PL * W_PLANE;            // 48 KB (40 KB with one plane, 24 KB with half-row A)
    constexpr int LPS = AR + PL * WR;                        // LDS-direct loads per wave per slice
    constexpr int RST = 3;                                   // ring stages (five 24 KB stages in the half-row mode measured no faster)
    __shared__ __attribute__((aligned(16))) char smem[RST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = RBN == 128 ? wave >> 1 : wave >> 2, wn = RBN == 128 ? wave & 1 : wave & 3;
    const int li = lane & 31, hi = lane >> 5;
    const int g8 = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int KT = p.K / BK;
    auto tile_of_round = [&](int r) { return (r * 8 + xcd) * g8 + slot; };
    if (tile_of_round(0) >= n_tiles) return;

    // ---- LDS-direct loader state (per lane) ----
    const int nw = (int)(((size_t)(p.N - 1) * p.ldw + p.K) * 2);
    const int na = AH ? (int)((size_t)(p.M - 1) * p.lda * 4 + (size_t)p.K * 2) : (int)(((size_t)(p.M - 1) * p.lda + p.K) * 4);
    const int arow = 8 * wave + (lane >> 3);                                  // row inside a 64-row instruction group
    const int wrow = 16 * wave + (lane >> 2);                                 // (wrow >> 2) & 3 == (lane >> 4) & 3
    const unsigned va = (AH && !LR) ? (unsigned)(wrow * p.lda * 4 + 16 * ((lane & 3) ^ ((lane >> 4) & 3)))
                                    : (unsigned)(arow * p.lda + 4 * ((lane & 7) ^ ((arow >> 1) & 7))) * 4u;
    const unsigned vw = LR ? (unsigned)(arow * p.ldw * 2 + 16 * ((lane & 7) ^ ((arow >> 1) & 7)))
                           : (unsigned)(wrow * p.ldw + 8 * ((lane & 3) ^ ((lane >> 4) & 3))) * 2u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, na, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.Whi), 0, nw, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(PL == 2 ? p.Wlo : p.Whi), 0, nw, 0x00020000);
    auto issue = [&](int m0, int n0, int k0, char* stage) {
        if (LR) {                     // 128-byte rows: 8 rows per instruction, 8 waves -> 64 rows per round
            char* sa = stage + wave * 8 * 128;
#pragma unroll
            for (int i = 0; i < AR; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 64 * 128, 16, va + (unsigned)((m0 + 64 * i) * p.lda * 4 + k0 * 2), 0, 0, 0);
            char* sw = stage + A_BYTES + wave * 8 * 128;
#pragma unroll
            for (int i = 0; i < WR; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, sw + i * 64 * 128, 16, vw + (unsigned)(((n0 + 64 * i) * p.ldw + k0) * 2), 0, 0, 0);
            return;
        }
        if (AH) {                     // 256 rows x 64 B: 16 rows per instruction, 8 waves -> 128 rows per round, two rounds
            char* sa = stage + wave * 16 * BK * 2;
#pragma unroll
            for (int i = 0; i < AR; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 128 * BK * 2, 16, va + (unsigned)((m0 + 128 * i) * p.lda * 4 + k0 * 2), 0, 0, 0);
        } else {
            float* sa = reinterpret_cast<float*>(stage) + wave * 8 * BK;
#pragma unroll
            for (int i = 0; i < AR; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 64 * BK, 16, va + (unsigned)(((m0 + 64 * i) * p.lda + k0) * 4), 0, 0, 0);
        }
        char* sw = stage + A_BYTES + wave * 16 * BK * 2;
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const unsigned off = vw + (unsigned)(((n0 + 128 * i) * p.ldw + k0) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, sw + i * 128 * BK * 2, 16, off, 0, 0, 0);
            if (PL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, sw + W_PLANE + i * 128 * BK * 2, 16, off, 0, 0, 0);
        }
    };

    // ---- issue cursor: two slices ahead of the compute cursor ----
    int ir = 0, ikt = 0, ibuf = 0, ahead = 0;                // round / k-slice / ring slot of the next slice to issue; slices in flight
    auto issue_next = [&]() {
        const int v = tile_of_round(ir);
        if (v >= n_tiles) return;
        if (!(kExperiments && (p.ablate & 1)) || (ir == 0 && ikt < RST)) issue((v / nbn) * RBM, (v % nbn) * RBN, ikt * BK, smem + ibuf * STAGE);
        ibuf = ibuf == RST - 1 ? 0 : ibuf + 1;
        if (++ikt == KT) { ikt = 0; ++ir; }
        ++ahead;
    };
#pragma unroll
    for (int i = 0; i < RST - 1; ++i) issue_next();

    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    int cbuf = 0;
    if constexpr (DB) {
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        struct Frags { bf16x8 a[4][TM], w[4][TN]; };
        const int swa = (li >> 1) & 7;
        auto read = [&](const char* stage, Frags& f) {
            const char* sAh = stage + (wm * 64 + li) * 128;
            const char* sW = stage + A_BYTES + (wn * 64 + li) * 128;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) f.a[ks][tm] = *reinterpret_cast<const bf16x8*>(sAh + tm * 32 * 128 + 16 * ((2 * ks + hi) ^ swa));
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) f.w[ks][tn] = *reinterpret_cast<const bf16x8*>(sW + tn * 32 * 128 + 16 * ((2 * ks + hi) ^ swa));
            }
        };
        auto mma = [&](const Frags& f) {
            if (kExperiments && (p.ablate & 2)) return;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[TM];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    s16x8 v = __builtin_bit_cast(s16x8, f.a[ks][tm]);
                    if (p.relu_a) v = __builtin_elementwise_max(v, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
                    a[tm] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = mfma_h<AFMT == 3>(f.w[ks][tn], a[tm], acc[tm][tn]);
            }
        };
        // one pipeline step: the slice that goes into `fill` has landed -> barrier (everyone is done reading the slice
        // in `use`, whose stage the next issue overwrites) -> issue -> start the reads into `fill` -> MFMAs on `use`
        auto step = [&](const Frags& use, Frags& fill, bool more) {
            if (more) {
                if (ahead >= RST - 1) wait_vmcnt<(RST - 2) * LPS>();
                else wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                --ahead;
                issue_next();
                read(smem + cbuf * STAGE, fill);
                cbuf = cbuf == RST - 1 ? 0 : cbuf + 1;
            }
            __builtin_amdgcn_sched_barrier(0);
            mma(use);
            __builtin_amdgcn_sched_barrier(0);
        };
        Frags f0, f1;
        {   // prologue: the first slice of the first tile into f0
            if (ahead >= RST - 1) wait_vmcnt<(RST - 2) * LPS>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            --ahead;
            issue_next();
            read(smem + cbuf * STAGE, f0);
            cbuf = cbuf == RST - 1 ? 0 : cbuf + 1;
        }
        for (int round = 0;; ++round) {
            const int v = tile_of_round(round);
            if (v >= n_tiles) break;
            const int m0 = (v / nbn) * RBM, n0 = (v % nbn) * RBN;
            const bool next_tile = tile_of_round(round + 1) < n_tiles;
            if (ADD != 0) tile_init<TM, TN, ADD>(p, m0, n0, wm, wn, lane, acc);
            for (int kt = 0; kt < KT; kt += 2) {                    // (KT is even: the launcher checks K % 128 == 0)
                step(f0, f1, true);
                step(f1, f0, kt + 2 < KT || next_tile);
            }
            tile_epilogue<TM, TN>(p, m0, n0, RBM, RBN, wm, wn, lane, acc);
            zero_acc<TM, TN>(acc);
        }
        return;
    }
    for (int round = 0;; ++round) {
        const int v = tile_of_round(round);
        if (v >= n_tiles) break;
        const int m0 = (v / nbn) * RBM, n0 = (v % nbn) * RBN;
        if (ADD != 0) tile_init<TM, TN, ADD>(p, m0, n0, wm, wn, lane, acc);      // additive operands straight into the accumulators
        for (int kt = 0; kt < KT; ++kt) {
            // this wave's part of the current slice has landed (at most one younger slice stays in flight) ...
            if (ahead >= RST - 1) wait_vmcnt<(RST - 2) * LPS>();
            else wait_vmcnt<0>();
            // ... everyone's has, and everyone is done with the slot the next issue overwrites
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            --ahead;
            issue_next();
            const char* stage = smem + cbuf * STAGE;
            cbuf = cbuf == RST - 1 ? 0 : cbuf + 1;
            const float* sA = reinterpret_cast<const float*>(stage) + (wm * 64 + li) * BK;
            const char* sAh = stage + (wm * 64 + li) * BK * 2;
            const char* sW = stage + A_BYTES + (wn * 64 + li) * BK * 2;
            const int swa = (li >> 1) & 7, sww = (li >> 2) & 3;
            auto slice = [&](auto relu_tag) {
                constexpr bool RELU = decltype(relu_tag)::value;
                constexpr int KS = BK / 16;
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                // ALL fragment reads of the slice first (both k-steps), then the MFMAs: k-step 0's matrix work runs while
                // k-step 1's fragments land, and the co-resident wave of the SIMD (same phase: one barrier per slice) finds
                // the LDS queue free sooner.  Left to itself hipcc emitted read -> wait -> MFMA batches, i.e. the LDS
                // latency several times per slice (ablation, kv launch: 405 us with the operand loads removed against a
                // 126 us MFMA bound).  The sched_barriers pin the order.
                f32x4 ax[KS][TM][2];
                bf16x8 ah[KS][TM], w[KS][PL][TN];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        if (LR) {
                            ah[ks][tm] = *reinterpret_cast<const bf16x8*>(sAh + tm * 32 * 128 + 16 * ((2 * ks + hi) ^ swa));
                        } else if (AH) {
                            ah[ks][tm] = *reinterpret_cast<const bf16x8*>(sAh + tm * 32 * BK * 2 + 16 * ((2 * ks + hi) ^ sww));
                        } else {
                            const int c0 = (4 * ks + 2 * hi) ^ swa;
                            ax[ks][tm][0] = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * c0);
                            ax[ks][tm][1] = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * (c0 ^ 1));
                        }
                    }
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            w[ks][pl][tn] = *reinterpret_cast<const bf16x8*>(sW + pl * W_PLANE + tn * 32 * BK * 2 + 16 * ((2 * ks + hi) ^ (LR ? swa : sww)));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 a[PL][TM];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        if (AH) {
                            s16x8 v = __builtin_bit_cast(s16x8, ah[ks][tm]);
                            if (RELU) v = __builtin_elementwise_max(v, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
                            a[0][tm] = __builtin_bit_cast(bf16x8, v);
                        } else {
                            Frag::template split8<RELU>(ax[ks][tm][0], ax[ks][tm][1], a[0][tm], a[PL - 1][tm]);
                        }
                    }
                    // term-major order: the four accumulators take turns (no MFMA waits for its predecessor)
                    if (PL == 2) {
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks][0][tn], a[1][tm], acc[tm][tn], 0, 0, 0);
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks][PL - 1][tn], a[0][tm], acc[tm][tn], 0, 0, 0);
                    }
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks][0][tn], a[0][tm], acc[tm][tn], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (kExperiments && (p.ablate & 2)) continue;                       // (timing experiment: no fragment reads, no MFMAs)
            if (p.relu_a) slice(std::true_type{});
            else slice(std::false_type{});
        }
        tile_epilogue<TM, TN>(p, m0, n0, RBM, RBN, wm, wn, lane, acc);
        zero_acc<TM, TN>(acc);
    }
}

}  // namespace

template <int T, int S, int ADD>
static void ring_launch(bool wide, const GemmArgs& a, int n_tiles, int nbn, int grid, hipStream_t s) {
    if constexpr (T == 1 && S >= 2) {          // half-row A, one plane: 64-wide slices whenever K allows
        if (!wide && a.K % 128 == 0 && !a.ring_bk32 && !a.ring_nodb) {
            hipLaunchKernelGGL((gemm_ring_kernel<T, S, ADD, 128, 64, true>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn);
            return;
        }
        if (!wide && a.K % 64 == 0 && !a.ring_bk32) {
            hipLaunchKernelGGL((gemm_ring_kernel<T, S, ADD, 128, 64>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn);
            return;
        }
    }
    if (!wide) hipLaunchKernelGGL((gemm_ring_kernel<T, S, ADD, 128>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn);
    else hipLaunchKernelGGL((gemm_ring_kernel<T, S, ADD, 256>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn);
}

// full rounds of a large-M bf16 launch; returns 1 if this operand combination is not built (the caller then uses the
// 128 x 128 kernel for everything)
int launch_gemm_ring(const GemmArgs& a, int rbn, int n_tiles, int grid, hipStream_t s) {
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
    if (a.rowscale || (add != 0 && add != 1 && add != 6)) return 1;
    const bool wide = rbn == 256;
    const int nbn = (a.N + rbn - 1) / rbn;
#define VLSAT_RING(T, S, ADD) ring_launch<T, S, ADD>(wide, a, n_tiles, nbn, grid, s)
#define VLSAT_RING_ADD(T, S)                      \
    switch (add) {                                \
        case 0: VLSAT_RING(T, S, 0); break;       \
        case 1: VLSAT_RING(T, S, 1); break;       \
        default: VLSAT_RING(T, S, 6); break;      \
    }
    if (a.prec == 3) {
        if (a.a_split == 2) return 1;
        if (a.a_split) { VLSAT_RING_ADD(3, 1) } else { VLSAT_RING_ADD(3, 0) }
    } else {
        if (a.a_split == 2 && a.half_f16) { VLSAT_RING_ADD(1, 3) } else if (a.a_split == 2) { VLSAT_RING_ADD(1, 2) } else if (a.a_split) { VLSAT_RING_ADD(1, 1) } else { VLSAT_RING_ADD(1, 0) }
    }
#undef VLSAT_RING_ADD
#undef VLSAT_RING
    if (a.launches) ++*a.launches;
    VLSAT_LAUNCH_CHECK("gemm_bf16_ring");
    return 0;
}

}  // namespace vlsat
