// (source only, not built: an experiment of round 2.  It was wired into launch_gemm for the split-bf16 full rounds and measured
//  against the 8-wave ring kernel on one box: identical launch times in tools/gemm_bench.py (kv 468 vs 466 us, nn_edge.2 439 vs
//  438 us) and 1 % fewer scenes/s end to end (3830 vs 3880), so it is not part of the library.  To try it again: copy it next
//  to gemm_bf16_ring.hip, add it to build.py and call launch_gemm_ring16 before launch_gemm_ring in launch_gemm.)
// Split-bf16 ring GEMM with SIXTEEN waves per CU (same contract, tile, ring and slice sequence as gemm_bf16_ring.hip).
//
// Why: tools/lds_bw_probe.hip -- with 8 waves on a CU (2 per SIMD) batched ds_read_b128 fragment reads reach ~60-65
// bytes per clock per CU, with 16 waves ~120-130: the LDS only gets near its 128 B/clk when four waves per SIMD keep reads
// in flight.  The 8-wave kernel reads 128 KB of fragments per 32-wide slice = ~2000 cycles at that rate, more than the
// 1536 cycles of MFMAs the slice carries (ablation in DESIGN.md section 8: the compute phase alone takes 2x its MFMA time).
// Here the 256 x 128 block tile is shared by 4 x 4 waves with 64 x 32 wave tiles: 32 accumulator registers per lane, six
// 16-byte fragment reads per six MFMAs and k-step (192 KB per slice, but at twice the rate), <= 128 VGPRs so that four
// waves fit a SIMD.  The weight planes are loaded by wave halves: waves 0-7 the hi plane, 8-15 the lo plane.
#include <type_traits>

#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

template <int N> __device__ __forceinline__ void wait_vmcnt16() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int AFMT, int ADD>
__global__ __launch_bounds__(1024, 4) void gemm_ring16_kernel(GemmArgs p, int n_tiles, int nbn) {
    static_assert(AFMT == 0 || AFMT == 1, "fp32 or split-pair A");
    using Frag = PipeSplitDma<128, 128, 3, AFMT>;            // split8
    constexpr int RBM = 256, RBN = 128, RST = 3, PL = 2;
    constexpr int TM = 2, TN = 1;
    constexpr int A_BYTES = RBM * BK * 4, W_PLANE = RBN * BK * 2;
    constexpr int STAGE = A_BYTES + PL * W_PLANE;            // 48 KB
    constexpr int LPS = 2 + 1;                               // per wave and slice: two A rounds (128 rows each), one weight-plane half
    __shared__ __attribute__((aligned(16))) char smem[RST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, hi = lane >> 5;
    const int g8 = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int KT = p.K / BK;
    auto tile_of_round = [&](int r) { return (r * 8 + xcd) * g8 + slot; };
    if (tile_of_round(0) >= n_tiles) return;

    // ---- LDS-direct loader state (per lane) ----
    const int nw = (int)(((size_t)(p.N - 1) * p.ldw + p.K) * 2);
    const int na = (int)(((size_t)(p.M - 1) * p.lda + p.K) * 4);
    const int arow = 8 * wave + (lane >> 3);                                  // row inside a 128-row instruction round
    const int wrow = 16 * (wave & 7) + (lane >> 2);                           // (wrow >> 2) & 3 == (lane >> 4) & 3
    const unsigned va = (unsigned)(arow * p.lda + 4 * ((lane & 7) ^ ((arow >> 1) & 7))) * 4u;
    const unsigned vw = (unsigned)(wrow * p.ldw + 8 * ((lane & 3) ^ ((lane >> 4) & 3))) * 2u;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, na, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wave < 8 ? p.Whi : p.Wlo), 0, nw, 0x00020000);
    auto issue = [&](int m0, int n0, int k0, char* stage) {
        float* sa = reinterpret_cast<float*>(stage) + wave * 8 * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 128 * BK, 16, va + (unsigned)(((m0 + 128 * i) * p.lda + k0) * 4), 0, 0, 0);
        char* sw = stage + A_BYTES + (wave >> 3) * W_PLANE + (wave & 7) * 16 * BK * 2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, sw, 16, vw + (unsigned)((n0 * p.ldw + k0) * 2), 0, 0, 0);
    };

    int ir = 0, ikt = 0, ibuf = 0, ahead = 0;
    auto issue_next = [&]() {
        const int v = tile_of_round(ir);
        if (v >= n_tiles) return;
        issue((v / nbn) * RBM, (v % nbn) * RBN, ikt * BK, smem + ibuf * STAGE);
        ibuf = ibuf == RST - 1 ? 0 : ibuf + 1;
        if (++ikt == KT) { ikt = 0; ++ir; }
        ++ahead;
    };
#pragma unroll
    for (int i = 0; i < RST - 1; ++i) issue_next();

    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);
    int cbuf = 0;
    for (int round = 0;; ++round) {
        const int v = tile_of_round(round);
        if (v >= n_tiles) break;
        const int m0 = (v / nbn) * RBM, n0 = (v % nbn) * RBN;
        if (ADD != 0) tile_init<TM, TN, ADD>(p, m0, n0, wm, wn, lane, acc);
        for (int kt = 0; kt < KT; ++kt) {
            if (ahead >= RST - 1) wait_vmcnt16<(RST - 2) * LPS>();
            else wait_vmcnt16<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            --ahead;
            issue_next();
            const char* stage = smem + cbuf * STAGE;
            cbuf = cbuf == RST - 1 ? 0 : cbuf + 1;
            const float* sA = reinterpret_cast<const float*>(stage) + (wm * 64 + li) * BK;
            const char* sW = stage + A_BYTES + (wn * 32 + li) * BK * 2;
            const int swa = (li >> 1) & 7, sww = (li >> 2) & 3;
            auto slice = [&](auto relu_tag) {
                constexpr bool RELU = decltype(relu_tag)::value;
                constexpr int KS = BK / 16;
                // fragments k-step by k-step (no second register set: 128 VGPRs per lane is the budget of four waves per SIMD;
                // the other three waves of the SIMD cover the LDS latency)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    f32x4 ax[TM][2];
                    bf16x8 w[PL];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        const int c0 = (4 * ks + 2 * hi) ^ swa;
                        ax[tm][0] = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * c0);
                        ax[tm][1] = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * (c0 ^ 1));
                    }
#pragma unroll
                    for (int pl = 0; pl < PL; ++pl)
                        w[pl] = *reinterpret_cast<const bf16x8*>(sW + pl * W_PLANE + 16 * ((2 * ks + hi) ^ sww));
                    bf16x8 a[PL][TM];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) Frag::template split8<RELU>(ax[tm][0], ax[tm][1], a[0][tm], a[1][tm]);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[1][tm], acc[tm][0], 0, 0, 0);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1], a[0][tm], acc[tm][0], 0, 0, 0);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], a[0][tm], acc[tm][0], 0, 0, 0);
                }
            };
            if (p.relu_a) slice(std::true_type{});
            else slice(std::false_type{});
        }
        tile_epilogue<TM, TN>(p, m0, n0, RBM, RBN, wm, wn, lane, acc);
        zero_acc<TM, TN>(acc);
    }
}

}  // namespace

// split-bf16 launches only (prec 3, A fp32 or split pairs); 1 = combination not built
int launch_gemm_ring16(const GemmArgs& a, int n_tiles, int grid, hipStream_t s) {
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
    if (a.prec != 3 || a.a_split == 2 || a.rowscale || (add != 0 && add != 1 && add != 6)) return 1;
    const int nbn = (a.N + 127) / 128;
#define VLSAT_R16(S, ADD) hipLaunchKernelGGL((gemm_ring16_kernel<S, ADD>), dim3(grid), dim3(1024), 0, s, a, n_tiles, nbn)
#define VLSAT_R16_ADD(S)                       \
    switch (add) {                             \
        case 0: VLSAT_R16(S, 0); break;        \
        case 1: VLSAT_R16(S, 1); break;        \
        default: VLSAT_R16(S, 6); break;       \
    }
    if (a.a_split) { VLSAT_R16_ADD(1) } else { VLSAT_R16_ADD(0) }
#undef VLSAT_R16_ADD
#undef VLSAT_R16
    if (a.launches) ++*a.launches;
    VLSAT_LAUNCH_CHECK("gemm_bf16_ring16");
    return 0;
}

}  // namespace vlsat
