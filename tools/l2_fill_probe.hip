// How fast can a CU pull L2-resident data into LDS (or VGPRs)?  The bf16 GEMMs of BASELINE configs[2] all plateau at
// ~29 GB/s per CU of operand traffic whatever the staging method; this probe measures the ceiling
// of that path without any arithmetic:
//   every block streams its own window (or, mode 'shared', the same window as every other block) `iters` times with
//   1 KiB-per-wave-instruction loads, `depth` instructions in flight per wave (counted vmcnt), 4 or 8 waves per block,
//   1 or 2 blocks per CU;  row length 128 B (an fp32 / split-pair k-slice) or 64 B (a bf16 weight-plane k-slice).
// Prints GB/s per CU and TB/s for the chip.
//   hipcc --offload-arch=gfx950 -O3 tools/l2_fill_probe.hip -o tools/bin/l2_fill_probe && tools/bin/l2_fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int MAXD = 12;
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void stream_kernel(const float* __restrict__ src, size_t window_floats, size_t block_stride_floats,
                                                     int row_bytes, int pitch_bytes, int iters, int depth, int to_lds,
                                                     float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) char lds[MAXD * 8 * 1024 + 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const float* base = src + (size_t)blockIdx.x * block_stride_floats;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(window_floats * 4), 0x00020000);
    // one instruction = 1 KiB = (1024 / row_bytes) rows of row_bytes; lane -> (row, 16-byte chunk)
    const int lanes_per_row = row_bytes / 16, rows_per_instr = 64 / lanes_per_row;
    const unsigned lane_off = (unsigned)((lane / lanes_per_row) * pitch_bytes + (lane % lanes_per_row) * 16);
    const unsigned instr_bytes = (unsigned)(rows_per_instr * pitch_bytes);
    const unsigned window = (unsigned)(window_floats * 4);
    float acc = 0.f;
    unsigned off = wave * instr_bytes;
    const unsigned step = nwaves * instr_bytes;
    f4 v[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) v[d] = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < MAXD; ++d) {
            if (d < depth) {
                const unsigned o = off + lane_off;
                if (to_lds) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds + (d * 8 + wave) * 1024, 16, o, 0, 0, 0);
                else v[d] = *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(base) + o);
                off += step;
                if (off + instr_bytes > window) off = wave * instr_bytes;
            }
        }
        if (to_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else {
#pragma unroll
            for (int d = 0; d < MAXD; ++d) acc += v[d][0];
        }
    }
    if (to_lds) acc = reinterpret_cast<float*>(lds)[tid];
    if (acc == 123.456f) sink[0] = acc;
}

template <int DEPTH, bool TO_LDS>
static void run(const char* what, const float* src, size_t total_floats, int blocks, int threads, bool shared, int row_bytes,
                int pitch_bytes, float* sink) {
    const size_t window = shared ? (1u << 18) : total_floats / blocks;            // shared: 1 MiB for everybody
    const size_t stride = shared ? 0 : window;
    const int iters = 400;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(threads), 0, 0, src, window, stride, row_bytes, pitch_bytes, iters, DEPTH, (int)TO_LDS, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * (threads / 64) * iters * DEPTH * 1024.0;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-58s %4d blocks x %d waves, depth %2d: %7.2f TB/s chip = %6.1f GB/s per CU (window %zu KiB per block)\n", what, blocks, threads / 64,
           DEPTH, tbs, tbs * 1e3 / 256, window * 4 / 1024);
}

int main() {
    const size_t total = (size_t)64 << 20;     // 256 MiB of floats?  no: 64 Mi floats = 256 MiB
    float *src = nullptr, *sink = nullptr;
    hipMalloc(&src, total * 4);
    hipMalloc(&sink, 64);
    hipMemset(src, 0, total * 4);
    const size_t totalbig = (size_t)1 << 30;    // 4 GiB
    float* big = nullptr;
    hipMalloc(&big, totalbig * 4);
    hipMemset(big, 0, totalbig * 4);
    // L2-resident windows: 256 blocks x 64 KiB = 16 MiB (2 MiB per XCD); 512 blocks x 32 KiB likewise
    const size_t small = (size_t)4 << 20;      // 16 MiB
    run<6, true>("LDS-direct, own 64 KiB window (L2 hits), contiguous", src, small, 256, 512, false, 128, 128, sink);
    run<12, true>("LDS-direct, own 64 KiB window (L2 hits), contiguous", src, small, 256, 512, false, 128, 128, sink);
    run<8, true>("LDS-direct, own 32 KiB window, 2 blocks/CU x 4 waves", src, small, 512, 256, false, 128, 128, sink);
    run<12, false>("to VGPRs, own 64 KiB window (L2 hits), contiguous", src, small, 256, 512, false, 128, 128, sink);
    run<6, true>("LDS-direct, SAME 1 MiB window for all blocks, 128 B rows", src, small, 256, 512, true, 128, 128, sink);
    run<6, true>("LDS-direct, SAME 1 MiB window for all blocks, 64 B rows", src, small, 256, 512, true, 64, 64, sink);
    run<12, false>("to VGPRs, SAME 1 MiB window for all blocks", src, small, 256, 512, true, 128, 128, sink);
    run<6, true>("LDS-direct, own 64 KiB window, 128 B of every 2 KiB row (4 KiB distinct)", src, small, 256, 512, false, 128, 2048, sink);
    run<6, true>("LDS-direct, own 1 MiB window, contiguous (HBM / Infinity Cache stream)", src, total, 256, 512, false, 128, 128, sink);
    run<12, true>("LDS-direct, own 1 MiB window, contiguous (HBM / Infinity Cache stream)", src, total, 256, 512, false, 128, 128, sink);
    run<12, true>("LDS-direct, own 1 MiB window, 128 B of every 2 KiB row (64 KiB distinct: L2)", src, total, 256, 512, false, 128, 2048, sink);
    run<12, true>("LDS-direct, own 16 MiB window, 128 B of every 2 KiB row (1 MiB distinct: beyond L2)", big, totalbig, 256, 512, false, 128, 2048, sink);
    run<12, true>("LDS-direct, own 16 MiB window, 256 B of every 2 KiB row", big, totalbig, 256, 512, false, 256, 2048, sink);
    run<12, true>("LDS-direct, own 16 MiB window, 512 B of every 2 KiB row", big, totalbig, 256, 512, false, 512, 2048, sink);
    run<12, true>("LDS-direct, own 16 MiB window, contiguous", big, totalbig, 256, 512, false, 128, 128, sink);
    run<12, false>("to VGPRs, own 1 MiB window, contiguous (HBM / Infinity Cache stream)", src, total, 256, 512, false, 128, 128, sink);
    return 0;
}
