// Probe: do VALU instructions hide behind v_mfma_f32_32x32x16_bf16 on gfx950, and how does it depend on how they are ordered
// in the wave and on the number of waves per SIMD?  Per iteration a wave issues 8 MFMAs (4 independent accumulators) and NV VALU
// instructions of kind KIND (0: v_add_f32 on 8 independent chains, 1: v_exp_f32 on 8 independent chains), either GROUPED
// (8 MFMAs, then the VALU block) or INTERLEAVED (NV / 8 VALU instructions behind every MFMA).  Registers only, no memory traffic.
// Reports shader cycles per iteration (s_memtime) next to the two lower bounds: MFMA pipe 8 x 32 = 256 cycles, VALU issue.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap && tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-result"
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
// KIND: 0 v_add_f32, 1 v_exp_f32, 2 v_pk_add_f32, 3 v_max3_f32, 4 v_cvt_pk_bf16_f32, 5 v_pk_mul_f32, 6 v_mul_f32 (round 5: which of the
// softmax's instructions share something with the matrix pipe?)
template <int KIND> __device__ __forceinline__ void valu(float& x, f32x2& y) {
    if (KIND == 0) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(x));
    else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(y));
    else if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x));
    else if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x));
    else if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(y));
    else asm volatile("v_mul_f32 %0, %0, %0" : "+v"(x));
}
template <int NV, int KIND, bool INTER>
__global__ __launch_bounds__(256, 2) void probe(const bf16x8* in, float* out, long long* cyc, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x16 c[4] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x * 1e-3f + i;
    f32x2 v2[4];
    for (int i = 0; i < 4; ++i) v2[i] = f32x2{v[i], v[i + 4]};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            c[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 3], b[(m >> 1) & 3], c[m & 3], 0, 0, 0);
            asm volatile("" : "+v"(c[m & 3]));
            if (INTER) {
#pragma unroll
                for (int k = 0; k < NV / 8; ++k) {
                    const int i = (m * (NV / 8) + k) & 7;
                    valu<KIND>(v[i], v2[i & 3]);
                }
            }
        }
        if (!INTER) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                valu<KIND>(v[k & 7], v2[k & 3]);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[0][r] + c[1][r] + c[2][r] + c[3][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += v2[i][0] + v2[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND, bool INTER>
static void run(const bf16x8* in, float* out, long long* cyc, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 4000;
    probe<NV, KIND, INTER><<<blocks, 256>>>(in, out, cyc, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<NV, KIND, INTER><<<blocks, 256>>>(in, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static long long h[2048];
    hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < blocks; ++i) avg += (double)h[i];
    avg /= blocks;
    // s_memtime counts at a constant 100 MHz on this part; convert with the wall time: cycles of the SIMD per iteration of ONE wave
    const double tf = (double)blocks * 4 * iters * 8 * 32768.0 / ms / 1e9;
    printf("%-5s %2d %s per 8 MFMAs, %s, %d wave(s)/SIMD: %7.3f ms, %6.1f TFLOP/s of MFMA, %6.1f ns per wave-iteration (MFMA pipe alone: %d waves x 256 cyc)\n",
           KIND == 0 ? "add" : KIND == 1 ? "exp" : KIND == 2 ? "pkadd" : KIND == 3 ? "max3" : KIND == 4 ? "cvtpk" : KIND == 5 ? "pkmul" : "mul", NV, "VALU", INTER ? "interleaved" : "grouped    ", waves_per_simd, ms, tf, ms * 1e6 / iters, waves_per_simd);
}

int main() {
    bf16x8* in; float* out; long long* cyc;
    hipMalloc(&in, 256 * 8 * sizeof(bf16x8));
    hipMalloc(&out, 2048 * 256 * 4);
    hipMalloc(&cyc, 2048 * sizeof(long long));
    __bf16 h[256 * 64];
    for (int i = 0; i < 256 * 64; ++i) h[i] = (__bf16)((float)rand() / RAND_MAX * 2.f - 1.f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run<0, 0, false>(in, out, cyc, w);
        run<32, 0, false>(in, out, cyc, w);
        run<32, 0, true>(in, out, cyc, w);
        run<64, 0, false>(in, out, cyc, w);
        run<64, 0, true>(in, out, cyc, w);
        run<32, 1, false>(in, out, cyc, w);
        run<32, 1, true>(in, out, cyc, w);
    }
    // round 5: instruction kinds of the softmax, 4 waves per SIMD, grouped (what the attention kernels do)
    run<64, 2, false>(in, out, cyc, 4);
    run<64, 3, false>(in, out, cyc, 4);
    run<64, 4, false>(in, out, cyc, 4);
    run<64, 5, false>(in, out, cyc, 4);
    run<64, 6, false>(in, out, cyc, 4);
    run<64, 2, true>(in, out, cyc, 4);
    run<64, 3, true>(in, out, cyc, 4);
    run<64, 4, true>(in, out, cyc, 4);
    return 0;
}
