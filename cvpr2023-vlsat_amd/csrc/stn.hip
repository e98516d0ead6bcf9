// Glue kernels of the MODEL.feature_transform path (STNkd, reference network_PointNet.py:52-86,146-150).
// With that switch every encoder becomes  h1 = relu(conv1 x);  T = STNkd(h1) [64x64 per object];  h1' = h1 . T;
// then conv2, conv3, max.  The dense layers of that chain run on the fp32 MFMA GEMM over point rows; these
// kernels provide what is not a GEMM: conv1 as point rows, the max over an object's points, and the per-object
// 64x64 transform.  The switch is off in the shipped configuration, so the path is written for correctness and
// bounded memory, not for the roofline.
#include "common.h"
#include "kernels.h"

namespace vlsat {

// rows[(n*P + p), c] = relu(b1[c] + sum_k w1[c,k] * pts[n, k, p]),  c < 64   (PointNetfeat conv1, :141-144)
__global__ __launch_bounds__(256) void pts_conv1_rows_kernel(const float* __restrict__ pts, int n_obj, int P, int cin,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             float* __restrict__ rows) {
    __shared__ float sW[64 * 10];
    for (int i = threadIdx.x; i < 64 * cin; i += 256) sW[(i / cin) * 10 + i % cin] = w1[i];
    __syncthreads();
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // (row, 16-channel quarter)
    const size_t row = idx >> 2;
    if (row >= (size_t)n_obj * P) return;
    const int q = (int)(idx & 3) * 16;
    const size_t n = row / P, p = row % P;
    float x[9];
    for (int k = 0; k < cin; ++k) x[k] = pts[(n * cin + k) * P + p];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ch = q + c4 * 4 + c;
            float a = b1[ch];
            for (int k = cin - 1; k >= 0; --k) a = fmaf(sW[ch * 10 + k], x[k], a);
            o[c] = fmaxf(a, 0.f);
        }
        *reinterpret_cast<f32x4*>(rows + row * 64 + q + c4 * 4) = o;
    }
}

int launch_pts_conv1_rows(const float* pts, int n_obj, int P, int cin, const float* w1, const float* b1, float* rows,
                          hipStream_t s) {
    if (n_obj <= 0) return 0;
    if (cin < 1 || cin > 9) return fail(-1, "pts_conv1_rows: 1..9 point channels");
    const size_t n4 = (size_t)n_obj * P * 4;
    hipLaunchKernelGGL(pts_conv1_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, pts, n_obj, P, cin, w1, b1, rows);
    VLSAT_LAUNCH_CHECK("pts_conv1_rows");
    return 0;
}

// out[n, c] = max_{p < P} x[(n*P + p), c]      (torch.max(x, 2) of PointNetfeat / STNkd)
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ x, int ld, int n_obj, int P, int cols,
                                                     float* __restrict__ out, int ldo) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = idx / cols;
    if (n >= (size_t)n_obj) return;
    const int c = (int)(idx % cols);
    const float* p = x + n * P * ld + c;
    float m = p[0];
    for (int i = 1; i < P; ++i) m = fmaxf(m, p[(size_t)i * ld]);
    out[n * ldo + c] = m;
}

int launch_rowmax(const float* x, int ld, int n_obj, int P, int cols, float* out, int ldo, hipStream_t s) {
    if (n_obj <= 0) return 0;
    const size_t n = (size_t)n_obj * cols;
    hipLaunchKernelGGL(rowmax_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ld, n_obj, P, cols, out, ldo);
    VLSAT_LAUNCH_CHECK("rowmax");
    return 0;
}

// out[r, j] = sum_i h[r, i] * T[r / P][i*64 + j]      (x^T . trans_feat, :148-150); one block of 64 threads per row
__global__ __launch_bounds__(256) void apply_stn_kernel(const float* __restrict__ h, int ldh, const float* __restrict__ T,
                                                        size_t rows, int P, float* __restrict__ out, int ldo) {
    __shared__ float sh[4][64];
    const int sub = threadIdx.x >> 6, j = threadIdx.x & 63;
    const size_t r = (size_t)blockIdx.x * 4 + sub;
    const bool ok = r < rows;
    sh[sub][j] = ok ? h[r * ldh + j] : 0.f;
    __syncthreads();
    if (!ok) return;
    const float* t = T + (r / P) * 4096 + j;
    float a = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) a = fmaf(sh[sub][i], t[i * 64], a);
    out[r * ldo + j] = a;
}

int launch_apply_stn(const float* h, int ldh, const float* T, size_t rows, int P, float* out, int ldo, hipStream_t s) {
    if (rows == 0) return 0;
    if (P < 1) return fail(-1, "apply_stn: P must be >= 1");
    hipLaunchKernelGGL(apply_stn_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, h, ldh, T, rows, P, out, ldo);
    VLSAT_LAUNCH_CHECK("apply_stn");
    return 0;
}

}  // namespace vlsat
