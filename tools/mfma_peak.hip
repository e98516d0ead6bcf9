// Ceiling probe: fp32 MFMA issue rate on this chip with random operands (power/clock limited),
// no memory traffic.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void probe(const float* in, float* out, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 8 + i]; }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[(i + 1) & 7], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 1) & 7], b[i], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + 3) & 7], b[(i + 5) & 7], c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int blocks = 512, iters = 20000;
    float *in, *out;
    hipMalloc(&in, 256 * 16 * 4);
    hipMalloc(&out, blocks * 256 * 4);
    float h[256 * 16];
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < 256 * 16; ++i) h[i] = zero ? 0.f : (float)rand() / RAND_MAX * 2e-3f - 1e-3f;
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<<<blocks, 256>>>(in, out, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        probe<<<blocks, 256>>>(in, out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 32 * 4096.0;
        printf("%s operands: %.2f ms, %.1f TFLOP/s fp32 MFMA (2 waves/SIMD)\n", zero ? "zero" : "random", ms, fl / ms / 1e9);
    }
    return 0;
}
