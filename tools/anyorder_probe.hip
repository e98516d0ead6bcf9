// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 / ROCm 7.2?  (hip_ext.h says "not supported on GFX9xx" for the
// module-launch form.)  Two spin kernels of 64 blocks x 100 us: 100 us total if they overlap, 200 us if the queue serialises them.
//   hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/bin/anyorder_probe && tools/bin/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int* d; hipMalloc(&d, 4);
    const long long ticks = 100 * 100;          // wall_clock64: 100 MHz
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(s);
            hipEventRecord(a, s);
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, ticks, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, ticks, d);
            else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
            else { hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
                   hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, ticks, d); }       // a third, ordered: must wait for both
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "two ordered" : mode == 1 ? "ordered + any-order" : "ordered + any-order + ordered", ms * 1e3);
        }
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
