// Probe: what does it cost a wave that owns its SIMD (one wave per SIMD, 64 fp32 MFMAs = 4096 pipe cycles per
// iteration) to issue NL 16-byte-per-lane global loads per iteration between those MFMAs?
//   variant 0: no loads                      variant 1: global_load_dwordx4 -> VGPRs
//   variant 2: global_load_lds_dwordx4       variant 3: raw buffer_load_dwordx4 -> VGPRs
//   variant 4: global_load_dwordx4 -> VGPRs followed one iteration later by ds_write_b128 (the GEMM's staging)
//   variant 5: buffer_load_dwordx4 ... lds   variant 6: buffer_load_dwordx4 -> VGPRs -> ds_write_b128
//   variant 7: ds_read_b128 (MFMA fragment reads, pitch-36 layout)      OCC = blocks (waves per SIMD) per CU
// Addresses follow the GEMM staging map (8 lanes per 128-byte line, rows 2 KB apart); `stream` = every block walks
// its own 768 KB region of a 200 MB buffer (HBM/MALL), otherwise all blocks re-read the same 768 KB (L2 hits).
//   hipcc --offload-arch=gfx950 -O3 tools/vmem_issue_probe.hip -o /tmp/vmem_probe && /tmp/vmem_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 384, KF = 512;            // one block's slab rows, floats per row

template <int VAR, int NL, int OCC>
__global__ __launch_bounds__(256, OCC) void probe(const float* __restrict__ src, float* out, long long* cyc, int iters, int stream) {
    __shared__ __attribute__((aligned(16))) float smem[30 * 1024 / OCC];    // 120 KB / OCC: OCC blocks per CU
    const int tid = threadIdx.x;
    const float* base = src + (stream ? (size_t)blockIdx.x * ROWS * KF : 0);
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 regs[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) regs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + tid * 1e-3f, b = 0.5f;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, ROWS * KF * 4, 0x00020000);
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int k0 = (it * 32) & (KF - 1);
        float* stage = smem + (it & 1) * (OCC == 1 ? 15 : 7) * 1024;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < NL) {
                const unsigned off = (unsigned)((srow + 32 * c) * KF + k0 + scol);
                if (VAR == 1) {
                    asm volatile("" :: "v"(regs[c]));                                   // consume last iteration's data
                    regs[c] = *reinterpret_cast<const f32x4*>(base + off);
                } else if (VAR == 4) {
                    *reinterpret_cast<f32x4*>(stage + (srow + 32 * c) * 36 + scol) = regs[c];
                    regs[c] = *reinterpret_cast<const f32x4*>(base + off);
                } else if (VAR == 2) {
                    __builtin_amdgcn_global_load_lds(base + off, stage + c * 1024 + (tid >> 6) * 256, 16, 0, 0);
                } else if (VAR == 3) {
                    asm volatile("" :: "v"(regs[c]));
                    regs[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off * 4), 0, 0));
                } else if (VAR == 5) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, stage + c * 1024 + (tid >> 6) * 256, 16, (int)(off * 4), 0, 0, 0);
                } else if (VAR == 7) {                                                   // fragment-style ds_read_b128
                    asm volatile("" :: "v"(regs[c]));
                    regs[c] = *reinterpret_cast<const f32x4*>(stage + ((tid & 31) + 32 * (c & 3)) * 36 + (c >> 2) * 8 + ((tid >> 5) & 1) * 4);
                } else if (VAR == 6) {
                    *reinterpret_cast<f32x4*>(stage + (srow + 32 * c) * 36 + scol) = regs[c];
                    regs[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off * 4), 0, 0));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = (c * 4 + q) & 7;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (VAR == 2 || VAR == 5) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NL > 0 ? NL : 0) : "memory");
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int j = 0; j < 12; ++j) s += regs[j][0];
    s += smem[tid];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR, int NL, int OCC = 1> void run(const float* src, float* out, long long* cyc, int stream, const char* name) {
    const int blocks = 256 * OCC, iters = 2000;
    hipLaunchKernelGGL((probe<VAR, NL, OCC>), dim3(blocks), dim3(256), 0, 0, src, out, cyc, 50, stream);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<VAR, NL, OCC>), dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters, stream);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    long long h[512];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)h[i];
    m /= blocks * (double)iters;
    const double ideal = 4096.0 * OCC;            // OCC waves share the SIMD
    printf("%-46s NL=%2d occ=%d %-7s %8.0f cycles/iter  (+%5.0f over %4.0f; %5.1f per op per wave)\n", name, NL, OCC, stream ? "stream" : "L2", m, m - ideal,
           ideal, NL ? (m - ideal) / NL / OCC : 0.0);
}

int main() {
    float *src, *out; long long* cyc;
    const size_t n = (size_t)256 * ROWS * KF;
    hipMalloc(&src, n * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 512 * 8);
    hipMemset(src, 0, n * 4);
    run<7, 4>(src, out, cyc, 0, "ds_read_b128");
    run<7, 8>(src, out, cyc, 0, "ds_read_b128");
    run<7, 16>(src, out, cyc, 0, "ds_read_b128");
    run<0, 0, 2>(src, out, cyc, 0, "no loads");
    run<7, 16, 2>(src, out, cyc, 0, "ds_read_b128");
    run<4, 8, 2>(src, out, cyc, 0, "global_load_dwordx4 -> VGPR -> ds_write_b128");
    run<5, 8, 2>(src, out, cyc, 0, "buffer_load_dwordx4 ... lds");
    for (int stream = 0; stream < 2; ++stream) {
        run<0, 0>(src, out, cyc, stream, "no loads");
        run<1, 6>(src, out, cyc, stream, "global_load_dwordx4 -> VGPR");
        run<1, 12>(src, out, cyc, stream, "global_load_dwordx4 -> VGPR");
        run<4, 6>(src, out, cyc, stream, "global_load_dwordx4 -> VGPR -> ds_write_b128");
        run<4, 12>(src, out, cyc, stream, "global_load_dwordx4 -> VGPR -> ds_write_b128");
        run<2, 6>(src, out, cyc, stream, "global_load_lds_dwordx4");
        run<2, 12>(src, out, cyc, stream, "global_load_lds_dwordx4");
        run<3, 6>(src, out, cyc, stream, "buffer_load_dwordx4 -> VGPR");
        run<3, 12>(src, out, cyc, stream, "buffer_load_dwordx4 -> VGPR");
        run<6, 6>(src, out, cyc, stream, "buffer_load_dwordx4 -> VGPR -> ds_write_b128");
        run<6, 12>(src, out, cyc, stream, "buffer_load_dwordx4 -> VGPR -> ds_write_b128");
        run<5, 6>(src, out, cyc, stream, "buffer_load_dwordx4 ... lds");
        run<5, 12>(src, out, cyc, stream, "buffer_load_dwordx4 ... lds");
    }
    return 0;
}
