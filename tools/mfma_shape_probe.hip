// Which fp32 MFMA shape is cheaper in power?  Same FLOPs, operands N(0,1)-like, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(const float* in, float* out, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 8 + i]; }
    long long t0 = clock64(), w0 = wall_clock64();
    float s = 0;
    if (SHAPE == 32) {
        f32x16 c[4] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + j) & 7], b[(i + 2 * j) & 7], c[j], 0, 0, 0);
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
    } else {
        f32x4 c[16] = {};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + j) & 7], b[(i + 3 * j) & 7], c[j], 0, 0, 0);
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += c[j][r];
    }
    if (blockIdx.x == 5 && threadIdx.x == 0 && iters > 1000)
        printf("   %.3f GHz shader clock\n", (double)(clock64() - t0) / ((double)(wall_clock64() - w0) / 100e6) / 1e9);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int SHAPE> void run(const float* in, float* out, const char* name) {
    const int blocks = 512, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<SHAPE><<<blocks, 256>>>(in, out, 500); hipDeviceSynchronize();
    hipEventRecord(e0); probe<SHAPE><<<blocks, 256>>>(in, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration per wave: 32 MFMA x 4096 flop (32x32x2)  or 64 MFMA x 2048 flop (16x16x4)
    double fl = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("%-12s %.2f ms  %.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
}
int main() {
    float *in, *out; hipMalloc(&in, 256 * 16 * 4); hipMalloc(&out, 512 * 256 * 4);
    float h[256 * 16];
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 256 * 16; ++i) { float u = 0; for (int k = 0; k < 12; ++k) u += (float)rand() / RAND_MAX; h[i] = (u - 6.f) * (rep ? 0.05f : 1.f); }
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        printf("-- operands ~N(0,%s)\n", rep ? "0.05" : "1");
        run<32>(in, out, "32x32x2"); run<16>(in, out, "16x16x4"); run<32>(in, out, "32x32x2"); run<16>(in, out, "16x16x4");
    }
    return 0;
}
