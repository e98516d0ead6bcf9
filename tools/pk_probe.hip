// probe: semantics of the inline-asm packed fp32 helpers of flash_attn_bf16.hip on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float max3f(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, float m) { f32x2 d; const f32x2 mm = {m, m}; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(mm)); return d; }
__global__ void k(const float* in, float* out) {
    const int t = threadIdx.x;
    f32x2 a = {in[4 * t], in[4 * t + 1]}, b = {in[4 * t + 2], in[4 * t + 3]};
    f32x2 s = pk_add(a, b), d = pk_sub(a, in[4 * t + 2]);
    out[6 * t] = s[0]; out[6 * t + 1] = s[1]; out[6 * t + 2] = d[0]; out[6 * t + 3] = d[1];
    out[6 * t + 4] = max3f(in[4 * t], in[4 * t + 1], in[4 * t + 2]);
    out[6 * t + 5] = max3f(in[4 * t + 3], -INFINITY, in[4 * t + 3]);
}
int main() {
    float h[256], o[384], *di, *dout;
    for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37) % 23) - 11.5f;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const float* x = h + 4 * t; const float* y = o + 6 * t;
        const float m3 = fmaxf(x[0], fmaxf(x[1], x[2]));
        if (y[0] != x[0] + x[2] || y[1] != x[1] + x[3] || y[2] != x[0] - x[2] || y[3] != x[1] - x[2] || y[4] != m3 || y[5] != x[3]) {
            if (bad < 4) printf("t %d: in %g %g %g %g -> add %g %g sub %g %g max3 %g %g\n", t, x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3], y[4], y[5]);
            ++bad;
        }
    }
    printf("pk probe: %d bad of 64\n", bad);
    return bad != 0;
}
