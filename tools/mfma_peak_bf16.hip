// Ceiling probe: v_mfma_f32_32x32x16_bf16 issue rate on this chip with random operands (power / clock limited), no memory
// traffic, 2 or 1 waves per SIMD, short and long runs (the clock drops under sustained matrix load).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o tools/bin/mfma_peak_bf16 && tools/bin/mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256, 2) void probe(const bf16x8* in, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[i], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + 1) & 3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + 1) & 3], b[i], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + 3) & 3], b[(i + 2) & 3], c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    bf16x8* in; float* out;
    hipMalloc(&in, 256 * 8 * sizeof(bf16x8));
    hipMalloc(&out, 1024 * 256 * 4);
    __bf16 h[256 * 64];
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < 256 * 64; ++i) h[i] = (__bf16)(zero ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f);
        hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        for (int blocks : {512, 256})               // 2 / 1 waves per SIMD (256 CUs, 4 waves per block)
            for (int iters : {2000, 200000}) {      // ~0.1 ms and ~10 ms of matrix work
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                probe<<<blocks, 256>>>(in, out, 100);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                probe<<<blocks, 256>>>(in, out, iters);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double fl = (double)blocks * 4 * iters * 16 * 32768.0;      // 4 waves x 16 MFMAs x 2*32*32*16 flop
                printf("%s operands, %d wave(s)/SIMD, %7.3f ms: %7.1f TFLOP/s (v_mfma_f32_32x32x16_bf16)\n", zero ? "zero  " : "random",
                       blocks / 256, ms, fl / ms / 1e9);
            }
    }
    return 0;
}
