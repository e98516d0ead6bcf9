// LDS read bandwidth of one CU with ds_read_b128, for the access patterns of the bf16 GEMM fragment reads: is the
// "compute phase" of those kernels (fragment reads + MFMAs) LDS-bound?  8 waves (512 threads), every lane issues
// `iters` x 16 reads of 16 bytes; cycles from s_memtime.  Prints bytes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_bw_probe.hip -o tools/bin/lds_bw_probe && tools/bin/lds_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

// mode 3: lane-linear ds_read_b64 (8 bytes per lane); mode 4: lane-linear ds_read_b32
// mode 0: lane-linear (lane * 16 bytes + k * 1 KiB)                       -- the best case
// mode 1: GEMM fragment pattern, 128-byte rows: row = lane & 31, chunk = (2k + (lane >> 5)) ^ ((row >> 1) & 7)
// mode 2: GEMM fragment pattern, 64-byte rows:  row = lane & 31, chunk = ((2k + (lane >> 5)) & 3) ^ ((row >> 2) & 3)
__global__ __launch_bounds__(1024) void lds_read_kernel(int mode, int iters, long long* cycles, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    const int row = lane & 31, hi = lane >> 5;
    const char* base = lds + (wave & 7) * 8192;                             // 8 KiB per wave: 64 rows x 128 B
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    float acc1 = 0.f;
    if (mode >= 3) {
        const long long t0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (mode == 3) {
                    const float2 v = *reinterpret_cast<const float2*>(base + (k & 15) * 512 + lane * 8);
                    asm volatile("" ::"v"(v));
                } else {
                    const float v = *reinterpret_cast<const float*>(base + (k & 15) * 256 + lane * 4);
                    asm volatile("" ::"v"(v));
                }
            }
        }
        const long long t1 = clock64();
        if (tid == 0) cycles[blockIdx.x] = t1 - t0;
        if (acc1 == 123.f) sink[0] = acc1;
        return;
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        f4 v[16];                                                               // sixteen reads in flight, consumed together
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const char* p;
            if (mode == 0) p = base + ((k & 7) * 1024 + lane * 16);
            else if (mode == 1) p = base + (row + 32 * (k & 1)) * 128 + 16 * (((2 * (k >> 2) + hi) & 7) ^ ((row >> 1) & 7));
            else p = base + (row + 32 * (k & 3)) * 64 + 16 * (((2 * (k >> 2) + hi) & 3) ^ ((row >> 2) & 3));
            v[k] = *reinterpret_cast<const f4*>(p);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
    }
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc[0] == 123.f) sink[0] = acc[1] + acc[2] + acc[3];
}

int main() {
    long long* d;
    float* sink;
    hipMalloc(&d, 8 * 256);
    hipMalloc(&sink, 64);
    const char* names[5] = {"ds_read_b128, lane-linear", "ds_read_b128, fragment pattern, 128-byte rows (XOR swizzle (row>>1)&7)",
                            "ds_read_b128, fragment pattern, 64-byte rows (XOR swizzle (row>>2)&3)", "ds_read_b64, lane-linear", "ds_read_b32, lane-linear"};
    for (int mode = 0; mode < 5; ++mode) {
        const int iters = 2000;
      for (int threads = 512; threads <= 1024; threads *= 2) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(lds_read_kernel, dim3(256), dim3(threads), 0, 0, mode, iters, d, sink);
        hipDeviceSynchronize();
        long long c[256];
        hipMemcpy(c, d, sizeof(c), hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < 256; ++i) mean += (double)c[i];
        mean /= 256;
        const double bytes = (double)threads * iters * 16 * (mode == 3 ? 8 : mode == 4 ? 4 : 16);
        printf("%-74s %2d waves: %7.1f bytes / clock / CU\n", names[mode], threads / 64, bytes / mean);
      }
    }
    return 0;
}
