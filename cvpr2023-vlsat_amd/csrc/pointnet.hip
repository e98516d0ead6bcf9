// Fused PointNet object encoder:  out[n, :] = max_p relu(W3 relu(W2 relu(W1 x_p + b1) + b2) + b3)
// = PointNetfeat.forward with global_feat=True, no STN, BatchNorm result discarded
// (reference src/model/model_utils/network_PointNet.py:141-164, built at SGFN_MMG/model.py:51-57).
// The reference runs three Conv1d(k=1) and writes a [N,768,P] fp32 intermediate before the max;
// here nothing but the [N,768] result touches HBM.
//
// One block (4 waves, 2x2) walks 64-point chunks of one object:
//   conv1 (CIN->64)  VALU, 16 outputs per thread, result H1[64][64] in LDS; CIN = 3 (xyz), 6 or 9 (MODEL.USE_RGB /
//                    USE_NORMAL add 3 channels each, reference SGFN_MMG/model.py:31-35)
//   conv2 (64->128)  fp32 MFMA, A = H1 from LDS, B = W2 streamed through the stage buffers;
//                    H2[64][128] = relu(.) overwrites H1's LDS (aliased)
//   conv3 (128->768) fp32 MFMA, 6 column chunks x 4 k-slices of W3 double-buffered in LDS,
//                    epilogue relu + column max over the 64 points kept as a RUNNING max in
//                    registers across chunks (in-lane over 16 rows + one cross-half shuffle)
// Points beyond P in the last chunk are clamped to the last point (duplicates do not change a max).
// When an object is split over several blocks (few objects, many points) the partial maxima
// are merged with integer atomicMax -- valid because post-ReLU values are >= 0.
// Roofline: fp32 MFMA (213 376 flop/point, 92 % of it conv3); L2->LDS weight traffic is
// 425 KB per 64-point chunk = 13.6 MFLOP, HBM traffic = the points once.
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

constexpr int PN_M = 64;      // points per chunk
constexpr int PN_P1 = 68;     // H1 pitch
constexpr int PN_P2 = 132;    // H2 pitch
constexpr int PN_MAXNC = 6;   // n_out / 128 <= 6

template <int CIN>
__global__ __launch_bounds__(256, 2) void pointnet_kernel(
    const float* __restrict__ pts, int P, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
    const float* __restrict__ b3, int n_out, float* __restrict__ out, int nsplit) {
    __shared__ __attribute__((aligned(16))) float sH[PN_M * PN_P2];          // H1 (pitch 68) then H2 (pitch 132)
    __shared__ __attribute__((aligned(16))) float sW[2 * 128 * LDT];         // weight slices [128][36] x 2
    constexpr int S1 = (CIN + 4) & ~3;                                        // (w_0 .. w_{CIN-1}, b) per channel, padded to x4
    __shared__ __attribute__((aligned(16))) float sW1[64 * S1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hi = lane >> 5;
    const int obj = blockIdx.x / nsplit, part = blockIdx.x % nsplit;
    const int n_chunks = (P + PN_M - 1) / PN_M;
    const int n_nc = n_out / 128;
    const float* op = pts + (size_t)obj * CIN * P;

    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) sW1[tid * S1 + c] = w1[tid * CIN + c];
        sW1[tid * S1 + CIN] = b1[tid];
    }
    float rmax[PN_MAXNC][2];
#pragma unroll
    for (int i = 0; i < PN_MAXNC; ++i) rmax[i][0] = rmax[i][1] = 0.f;
    __syncthreads();

    f32x4 rw[4];
    for (int ch = part; ch < n_chunks; ch += nsplit) {
        // ---- conv1: thread -> point pp, channels cg..cg+15 ----
        {
            const int pp = tid >> 2, cg = (tid & 3) * 16;
            int p = ch * PN_M + pp;
            p = p < P ? p : P - 1;
            float xin[CIN];
#pragma unroll
            for (int c = 0; c < CIN; ++c) xin[c] = op[(size_t)c * P + p];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                f32x4 h;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* w = sW1 + (cg + c4 * 4 + c) * S1;
                    float a = w[CIN];                                    // b + w_{CIN-1} x_{CIN-1} + ... + w_0 x_0
#pragma unroll
                    for (int k = CIN - 1; k >= 0; --k) a = fmaf(w[k], xin[k], a);
                    h[c] = fmaxf(a, 0.f);
                }
                *reinterpret_cast<f32x4*>(sH + pp * PN_P1 + cg + c4 * 4) = h;
            }
        }
        // ---- conv2: both k-slices of W2 [128][64] into the two stage buffers ----
        stage_load<128>(w2, 64, 0, 127, 0, rw, tid);
        stage_store<128>(sW, rw, tid);
        stage_load<128>(w2, 64, 0, 127, 32, rw, tid);
        stage_store<128>(sW + 128 * LDT, rw, tid);
        __syncthreads();
        f32x16 acc[1][2];
        zero_acc<1, 2>(acc);
        mma_slice<1, 2, PN_P1, LDT>(sH + (wm * 32) * PN_P1, sW + (wn * 64) * LDT, acc, lane);
        mma_slice<1, 2, PN_P1, LDT>(sH + (wm * 32) * PN_P1 + 32, sW + 128 * LDT + (wn * 64) * LDT, acc, lane);
        stage_load<128>(w3, 128, 0, n_out - 1, 0, rw, tid);          // first W3 slice, in flight over the barrier
        __syncthreads();                                              // everyone done reading H1 / W2
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = wn * 64 + tn * 32 + li;
            const float bb = b2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sH[(wm * 32 + crow32(r, hi)) * PN_P2 + col] = fmaxf(acc[0][tn][r] + bb, 0.f);
        }
        stage_store<128>(sW, rw, tid);
        __syncthreads();
        // ---- conv3: 6 column chunks x 4 k-slices ----
        const int n_slices = n_nc * 4;
        zero_acc<1, 2>(acc);
        for (int j = 0; j < n_slices; ++j) {
            const int nc = j >> 2, ks = j & 3;
            const bool more = j + 1 < n_slices;
            if (more) stage_load<128>(w3, 128, ((j + 1) >> 2) * 128, n_out - 1, ((j + 1) & 3) * 32, rw, tid);
            mma_slice<1, 2, PN_P2, LDT>(sH + (wm * 32) * PN_P2 + ks * 32, sW + (j & 1) * 128 * LDT + (wn * 64) * LDT,
                                        acc, lane);
            if (ks == 3) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const float bb = b3[nc * 128 + wn * 64 + tn * 32 + li];
                    float m = 0.f;                                   // relu folded into the max with 0
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[0][tn][r] + bb);
                    m = half_max(m);
#pragma unroll
                    for (int q = 0; q < PN_MAXNC; ++q)               // static register index
                        if (q == nc) rmax[q][tn] = fmaxf(rmax[q][tn], m);
                }
                zero_acc<1, 2>(acc);
            }
            if (more) stage_store<128>(sW + ((j + 1) & 1) * 128 * LDT, rw, tid);
            __syncthreads();
        }
    }
    // ---- merge the two row-halves (wm) and write / atomically merge ----
    float* sR = sH;                                                   // [2][768]
    if (hi == 0) {
#pragma unroll
        for (int q = 0; q < PN_MAXNC; ++q)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                if (q < n_nc) sR[wm * 768 + q * 128 + wn * 64 + tn * 32 + li] = rmax[q][tn];
    }
    __syncthreads();
    for (int c = tid; c < n_out; c += 256) {
        const float v = fmaxf(sR[c], sR[768 + c]);
        float* dst = out + (size_t)obj * n_out + c;
        if (nsplit == 1) *dst = v;
        else atomicMax(reinterpret_cast<int*>(dst), __float_as_int(v));
    }
}

int launch_pointnet(const float* pts, int n_obj, int n_points, int cin, const float* w1, const float* b1,
                    const float* w2, const float* b2, const float* w3, const float* b3, int n_out,
                    float* out, hipStream_t s) {
    if (n_obj <= 0) return 0;
    if (n_points <= 0) return fail(-1, "pointnet: n_points must be > 0");
    if (n_out % 128 || n_out > 128 * PN_MAXNC) return fail(-1, "pointnet: n_out must be a multiple of 128, <= 768");
    const int n_chunks = (n_points + PN_M - 1) / PN_M;
    int nsplit = (512 + n_obj - 1) / n_obj;
    if (nsplit > n_chunks) nsplit = n_chunks;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 1 && launch_zero_f32(out, (size_t)n_obj * n_out, s)) return -1;
#define VLSAT_PN_CASE(CIN) \
    case CIN: hipLaunchKernelGGL(pointnet_kernel<CIN>, dim3(n_obj * nsplit), dim3(256), 0, s, pts, n_points, w1, b1, w2, \
                                 b2, w3, b3, n_out, out, nsplit); break;
    switch (cin) {
        VLSAT_PN_CASE(3) VLSAT_PN_CASE(6) VLSAT_PN_CASE(9)
        default: return fail(-1, "pointnet: point channels must be 3, 6 or 9");
    }
#undef VLSAT_PN_CASE
    VLSAT_LAUNCH_CHECK("pointnet");
    return 0;
}

}  // namespace vlsat
