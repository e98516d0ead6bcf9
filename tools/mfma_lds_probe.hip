// Probe: the GEMM inner loop (ds_read_b128 fragments -> 32x32x2 fp32 MFMAs) without any global
// traffic or barriers, to find what limits MFMA issue.  Variants selected by argv[1].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../cvpr2023-vlsat_amd/csrc/gemm_core.h"
namespace vlsat { void set_error(const std::string&) {} int fail(int c, const std::string&) { return c; } }
using namespace vlsat;

template <int VAR>
__global__ __launch_bounds__(256, 2) void probe(const float* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 256 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 256 * LDT; i += 256) smem[i] = in[i % 4096];
    __syncthreads();
    f32x16 acc[2][2];
    zero_acc<2, 2>(acc);
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const float* cur = smem + (it & 1) * 256 * LDT;
        if (VAR == 0) {
            mma_slice<2, 2>(cur + (wm * 64) * LDT, cur + 128 * LDT + (wn * 64) * LDT, acc, lane);
        } else if (VAR == 1) {   // all 16 fragment reads up front, then 64 MFMAs
            const int li = lane & 31, hi = lane >> 5;
            f32x4 a[4][2], b[4][2];
#pragma unroll
            for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[kg][t] = *reinterpret_cast<const f32x4*>(cur + (wm * 64 + t * 32 + li) * LDT + kg * 8 + hi * 4);
                    b[kg][t] = *reinterpret_cast<const f32x4*>(cur + 128 * LDT + (wn * 64 + t * 32 + li) * LDT + kg * 8 + hi * 4);
                }
#pragma unroll
            for (int kg = 0; kg < 4; ++kg)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg][tm][s], b[kg][tn][s], acc[tm][tn], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    if (blockIdx.x == 7 && tid == 0 && iters > 1000)
        printf("   block 7: %.3f GHz shader clock\n", (double)(clock64() - t0) / ((double)(wall_clock64() - w0) / 100e6) / 1e9);
    out[blockIdx.x * 256 + tid] = s;
}

template <int VAR> void run(const float* in, float* out, const char* name) {
    const int blocks = 512, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<VAR><<<blocks, 256>>>(in, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<VAR><<<blocks, 256>>>(in, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 64 * 4096.0;
    printf("%-40s %.2f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}
int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 512 * 256 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2e-2f - 1e-2f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(in, out, "mma_slice (2 sets, sched_barrier)");
    run<1>(in, out, "16 reads up front then 64 MFMA");
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("-- operands uniform(-2,2)\n");
    run<0>(in, out, "mma_slice (2 sets, sched_barrier)");
    run<1>(in, out, "16 reads up front then 64 MFMA");
    return 0;
}
