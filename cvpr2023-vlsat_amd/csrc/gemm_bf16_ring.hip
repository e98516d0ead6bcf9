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

constexpr int RBM = 256, RBN = 128, RST = 3;

template <int TERMS, int AFMT, int ADD>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(GemmArgs p, int n_tiles, int nbn) {
    using Frag = PipeSplitDma<128, 128, TERMS, AFMT>;        // fragment-side helpers only (split8 / frag_half)
    constexpr bool AH = AFMT == 2;                           // A as half rows (bf16): 64-byte slices like the weight planes
    constexpr int PL = TERMS == 1 ? 1 : 2;
    constexpr int TM = 2, TN = 2;
    constexpr int A_BYTES = AH ? RBM * BK * 2 : RBM * BK * 4, W_PLANE = RBN * BK * 2;
    constexpr int STAGE = A_BYTES + PL * W_PLANE;            // 48 KB (40 KB with one plane, 24 KB with half-row A)
    constexpr int LPS = (AH ? 2 : 4) + PL;                   // LDS-direct loads per wave per slice
    __shared__ __attribute__((aligned(16))) char smem[RST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
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
    const unsigned va = AH ? (unsigned)(wrow * p.lda * 4 + 16 * ((lane & 3) ^ ((lane >> 4) & 3)))
                           : (unsigned)(arow * p.lda + 4 * ((lane & 7) ^ ((arow >> 1) & 7))) * 4u;
    const unsigned vw = (unsigned)(wrow * p.ldw + 8 * ((lane & 3) ^ ((lane >> 4) & 3))) * 2u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, na, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwh = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.Whi), 0, nw, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(PL == 2 ? p.Wlo : p.Whi), 0, nw, 0x00020000);
    auto issue = [&](int m0, int n0, int k0, char* stage) {
        if (AH) {                     // 256 rows x 64 B: 16 rows per instruction, 8 waves -> 128 rows per round, two rounds
            char* sa = stage + wave * 16 * BK * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 128 * BK * 2, 16, va + (unsigned)((m0 + 128 * i) * p.lda * 4 + k0 * 2), 0, 0, 0);
        } else {
            float* sa = reinterpret_cast<float*>(stage) + wave * 8 * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 64 * BK, 16, va + (unsigned)(((m0 + 64 * i) * p.lda + k0) * 4), 0, 0, 0);
        }
        char* sw = stage + A_BYTES + wave * 16 * BK * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwh, sw, 16, vw + (unsigned)((n0 * p.ldw + k0) * 2), 0, 0, 0);
        if (PL == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, sw + W_PLANE, 16, vw + (unsigned)((n0 * p.ldw + k0) * 2), 0, 0, 0);
    };

    // ---- issue cursor: two slices ahead of the compute cursor ----
    int ir = 0, ikt = 0, ibuf = 0, ahead = 0;                // round / k-slice / ring slot of the next slice to issue; slices in flight
    auto issue_next = [&]() {
        const int v = tile_of_round(ir);
        if (v >= n_tiles) return;
        issue((v / nbn) * RBM, (v % nbn) * RBN, ikt * BK, smem + ibuf * STAGE);
        ibuf = ibuf == RST - 1 ? 0 : ibuf + 1;
        if (++ikt == KT) { ikt = 0; ++ir; }
        ++ahead;
    };
    issue_next();
    issue_next();

    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    int cbuf = 0;
    for (int round = 0;; ++round) {
        const int v = tile_of_round(round);
        if (v >= n_tiles) break;
        const int m0 = (v / nbn) * RBM, n0 = (v % nbn) * RBN;
        if (ADD != 0) {
            // additive operands (residual / gathered rows) straight into the accumulators (see gemm_f32.hip)
            int ldr = p.ldr, ldg0 = p.ldg0, ldg1 = p.ldg1, lv = lane;
            asm volatile("" : "+s"(ldr), "+s"(ldg0), "+s"(ldg1), "+v"(lv));
            const int l2 = lv & 31, h2 = lv >> 5;
            const float* rbase = (ADD & 1) ? p.resid + (size_t)m0 * ldr + n0 : nullptr;
with_resid_format((ADD & 1) ? p.r_split : 0, [&](auto fmt) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                int nl = (wn * TN + tn) * 32 + l2;
                if (n0 + nl >= p.N) nl = p.N - 1 - n0;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int ml = (wm * TM + tm) * 32 + crow32(r, h2);
                        if (m0 + ml >= p.M) ml = p.M - 1 - m0;
                        float x = 0.f;
                        if (ADD & 1) {
                            x = p.resid_scale * load_resid<decltype(fmt)::value>(rbase, ml, nl, ldr, n0);
                        }
                        if (ADD & 2) x += p.g0[(unsigned)(p.gi0[m0 + ml] * ldg0 + n0 + nl)];
                        if (ADD & 4) x += p.g1[(unsigned)(p.gi1[m0 + ml] * ldg1 + n0 + nl)];
                        acc[tm][tn][r] = x;
                    }
            }
            });
        }
        for (int kt = 0; kt < KT; ++kt) {
            // this wave's part of the current slice has landed (at most one younger slice stays in flight) ...
            if (ahead >= 2) {
                if (LPS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (LPS == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
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
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    bf16x8 a[PL][TM], w[PL][TN];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        if (AH) {
                            a[0][tm] = Frag::template frag_half<RELU>(sAh + tm * 32 * BK * 2 + 16 * ((2 * ks + hi) ^ sww));
                        } else {
                            const int c0 = (4 * ks + 2 * hi) ^ swa;
                            const f32x4 x0 = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * c0);
                            const f32x4 x1 = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * (c0 ^ 1));
                            Frag::template split8<RELU>(x0, x1, a[0][tm], a[PL - 1][tm]);
                        }
                    }
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            w[pl][tn] = *reinterpret_cast<const bf16x8*>(sW + pl * W_PLANE + tn * 32 * BK * 2 + 16 * ((2 * ks + hi) ^ sww));
                    // term-major order: the four accumulators take turns (no MFMA waits for its predecessor)
                    if (PL == 2) {
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][tm], w[0][tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tm], w[PL - 1][tn], acc[tm][tn], 0, 0, 0);
                    }
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][tm], w[0][tn], acc[tm][tn], 0, 0, 0);
                }
            };
            if (p.relu_a) slice(std::true_type{});
            else slice(std::false_type{});
        }
        // ---- epilogue (as gemm_f32.hip: straight-line code under wave-uniform flags) ----
        int ldc = p.ldc, lv = lane;
        asm volatile("" : "+s"(ldc), "+v"(lv));
        const int l2 = lv & 31, h2 = lv >> 5;
        float* cbase = p.C + (size_t)m0 * ldc + n0;
        if (p.bias) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                int n = n0 + (wn * TN + tn) * 32 + l2;
                n = n < p.N ? n : p.N - 1;
                const float bn = p.bias[n];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] += bn;
            }
        }
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] = fmaxf(acc[tm][tn][r], 0.f);
        } else if (p.act == ACT_SIGMOID) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 1.f / (1.f + __expf(-acc[tm][tn][r]));
        }
        if (p.c_scale != 1.f) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= p.c_scale;
        }
        if (p.c_split == 2) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = (wm * TM + tm) * 32 + crow32(r, h2), nl = (wn * TN + tn) * 32 + l2;
                        if (m0 + ml < p.M && n0 + nl < p.N) store_half(cbase, ml, nl, ldc, n0, acc[tm][tn][r]);
                    }
        } else {
        if (p.c_split) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] = pack_split(acc[tm][tn][r]);
        }
        if (m0 + RBM <= p.M && n0 + RBN <= p.N) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = (wm * TM + tm) * 32 + crow32(r, h2), nl = (wn * TN + tn) * 32 + l2;
                        cbase[(unsigned)(ml * ldc + nl)] = acc[tm][tn][r];
                    }
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = (wm * TM + tm) * 32 + crow32(r, h2), nl = (wn * TN + tn) * 32 + l2;
                        if (m0 + ml < p.M && n0 + nl < p.N) cbase[(unsigned)(ml * ldc + nl)] = acc[tm][tn][r];
                    }
        }
        }
        zero_acc<TM, TN>(acc);
    }
}

}  // namespace

// full rounds of a large-M bf16 launch; returns 1 if this operand combination is not built (the caller then uses the
// 128 x 128 kernel for everything)
int launch_gemm_ring(const GemmArgs& a, int n_tiles, int grid, hipStream_t s) {
    const int nbn = (a.N + RBN - 1) / RBN;
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
    if (a.rowscale || (add != 0 && add != 1 && add != 6)) return 1;
#define VLSAT_RING(T, S, ADD) hipLaunchKernelGGL((gemm_ring_kernel<T, S, ADD>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn)
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
        if (a.a_split == 2) { VLSAT_RING_ADD(1, 2) } else if (a.a_split) { VLSAT_RING_ADD(1, 1) } else { VLSAT_RING_ADD(1, 0) }
    }
#undef VLSAT_RING_ADD
#undef VLSAT_RING
    if (a.launches) ++*a.launches;
    VLSAT_LAUNCH_CHECK("gemm_bf16_ring");
    return 0;
}

}  // namespace vlsat
