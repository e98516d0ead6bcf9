// fp32 MFMA GEMM with fused epilogues -- the engine behind every nn.Linear / Conv1d(k=1) of
// the path (reference call sites: network_MMG.py:40,93-98; attention.py:54-58,77;
// network_PointNet.py:329-339; SGFN_MMG/model.py:294,305-306,329-330; clip_adapter/model.py:27-29).
//
// C[M,N] = act(rowscale * (A[M,K] . W[N,K]^T) + bias + resid_scale*resid + g0[gi0] + g1[gi1])
//
// Block = 256 threads = 4 waves in a 2x2 grid; block tile BM x BN, wave tile (BM/2) x (BN/2)
// made of 32x32 MFMA tiles; K streamed in BK=32 slices through double-buffered LDS with
// register staging (global loads of slice t+1 are issued before the MFMAs of slice t and
// written to the other buffer afterwards; one barrier per slice).
// Roofline: fp32 MFMA (157 TF chip peak).  Per 128x128x32 slice a block moves 32 KB from
// L2 for 1.05 MFLOP, ~19 GB/s/CU at peak rate, far inside L2 bandwidth, so the kernel is
// MFMA-issue bound; LDS traffic is 4 ds_read_b128 per 16 MFMAs.
// Grid is 1-D with an XCD-aware remap: all N-tiles of an M-panel run back-to-back on ONE XCD,
// so the A panel is fetched from HBM once and re-used out of that XCD's L2.
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int STAGE = (BM + BN) * LDT;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbn = (p.N + BN - 1) / BN;
    const int nbm = (p.M + BM - 1) / BM;
    const int v = xcd_remap(blockIdx.x, nbm * nbn);
    const int m0 = (v / nbn) * BM, n0 = (v % nbn) * BN;

    f32x16 acc[TM][TN];
    zero_acc<TM, TN>(acc);

    f32x4 ra[BM / 32], rb[BN / 32];
    const int KT = p.K / BK;

    stage_load<BM>(p.A, p.lda, m0, p.M - 1, 0, ra, tid);
    stage_load<BN>(p.W, p.ldw, n0, p.N - 1, 0, rb, tid);
    if (p.relu_a) stage_relu<BM>(ra);
    stage_store<BM>(smem, ra, tid);
    stage_store<BN>(smem + BM * LDT, rb, tid);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        float* cur = smem + (kt & 1) * STAGE;
        float* nxt = smem + ((kt + 1) & 1) * STAGE;
        const bool more = kt + 1 < KT;
        if (more) {
            stage_load<BM>(p.A, p.lda, m0, p.M - 1, (kt + 1) * BK, ra, tid);
            stage_load<BN>(p.W, p.ldw, n0, p.N - 1, (kt + 1) * BK, rb, tid);
        }
        mma_slice<TM, TN>(cur + (wm * TM * 32) * LDT, cur + BM * LDT + (wn * TN * 32) * LDT, acc, lane);
        if (more) {
            if (p.relu_a) stage_relu<BM>(ra);
            stage_store<BM>(nxt, ra, tid);
            stage_store<BN>(nxt + BM * LDT, rb, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds column n, 16 rows per 32x32 tile ----
    const int li = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + li;
        if (n >= p.N) continue;
        const float bn = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + tm) * 32 + crow32(r, hi);
                if (m >= p.M) continue;
                float x = acc[tm][tn][r];
                if (p.rowscale) x *= p.rowscale[m];
                x += bn;
                if (p.resid) x += p.resid_scale * p.resid[(size_t)m * p.ldr + n];
                if (p.g0) x += p.g0[(size_t)p.gi0[m] * p.ldg0 + n];
                if (p.g1) x += p.g1[(size_t)p.gi1[m] * p.ldg1 + n];
                if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
                else if (p.act == ACT_SIGMOID) x = 1.f / (1.f + __expf(-x));
                p.C[(size_t)m * p.ldc + n] = x;
            }
        }
    }
}

double gemm_flops(const GemmArgs& a) { return 2.0 * a.M * (double)a.N * a.K; }

template <int BM, int BN>
static int launch_t(const GemmArgs& a, hipStream_t s) {
    const int nb = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN>), dim3(nb), dim3(256), 0, s, a);
    VLSAT_LAUNCH_CHECK("gemm_f32");
    return 0;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    if (!a.A || !a.W || !a.C) return fail(-1, "gemm: null A/W/C");
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K <= 0 || a.K % BK) return fail(-1, "gemm: K must be a positive multiple of 32");
    if ((a.lda & 3) || (a.ldw & 3)) return fail(-1, "gemm: lda/ldw must be multiples of 4 floats");
    if ((reinterpret_cast<uintptr_t>(a.A) & 15) || (reinterpret_cast<uintptr_t>(a.W) & 15))
        return fail(-1, "gemm: A/W must be 16-byte aligned");
    auto blocks = [&](int bm, int bn) { return (long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
    // Largest tile that still gives every CU (256) two blocks; small problems take the
    // smallest tile so the launch covers as many CUs as the problem allows.
    if (a.N > 64 && blocks(128, 128) >= 512) return launch_t<128, 128>(a, s);
    if (a.N <= 64 && blocks(128, 64) >= 512) return launch_t<128, 64>(a, s);
    if (a.N > 64 && blocks(64, 128) >= 512) return launch_t<64, 128>(a, s);
    return launch_t<64, 64>(a, s);
}

}  // namespace vlsat
