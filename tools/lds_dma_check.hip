// Semantics check for buffer_load_dwordx4 ... lds on gfx950: where does lane l's 16 bytes land, and what do
// out-of-range offsets return?   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_check.hip -o /tmp/ldsdma && /tmp/ldsdma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, int n_floats, float* out) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 256 + 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 4 * 256 + 64; i += 256) smem[i] = -1.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_floats * 4, 0x00020000);
    // lane l fetches float4 number (63 - l) of this wave's 256-float source segment; the last wave runs off the end
    const int voff = (wave * 256 + (63 - lane) * 4) * 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, smem + wave * 256, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4 * 256 + 64; i += 256) out[i] = smem[i];
}
int main() {
    const int n = 3 * 256 + 128;          // the 4th wave's upper half is out of range
    float h[4 * 256], *d, *o, r[4 * 256 + 64];
    for (int i = 0; i < 4 * 256; ++i) h[i] = (float)i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 256>>>(d, n, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int c = 0; c < 4; ++c) {
                const int srcidx = w * 256 + (63 - l) * 4 + c;
                const float expect = srcidx < n ? (float)srcidx : 0.f;
                const float got = r[w * 256 + l * 4 + c];
                if (got != expect && bad++ < 8) printf("wave %d lane %d c %d: got %g expect %g\n", w, l, c, got, expect);
            }
    for (int i = 0; i < 64; ++i) if (r[1024 + i] != -1.f) { ++bad; printf("guard %d overwritten: %g\n", i, r[1024 + i]); }
    printf(bad ? "MISMATCH (%d)\n" : "OK: lane l -> LDS base + 16*l, out-of-range reads give 0\n", bad);
    return 0;
}
