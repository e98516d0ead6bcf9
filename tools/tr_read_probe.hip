// Semantics of ds_read_b64_tr_b16 on gfx950 (the LDS transpose read the split-bf16 attention uses for its V operand).
//
// Model under test (cdna_hip_programming.md section 2 / T10, restated): within each group of 16 consecutive lanes, lane
// i supplies the address of an 8-byte piece (4 bf16); the 16 pieces form a 4 x 16 matrix, piece i = row i>>2, columns
// 4(i&3)..+3; lane i receives column i: element j of its result = element (i&3) of the piece addressed by lane
// 4j + (i>>2) of its group.  The program checks the model with (a) a contiguous image, (b) arbitrary per-lane row
// addresses (free row stride), (c) every lane on the same address, and prints what the hardware returned where the
// model fails.
//
//   hipcc --offload-arch=gfx950 -O3 tools/tr_read_probe.hip -o tools/bin/tr_read_probe && tools/bin/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int* __restrict__ addr_elems, short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;        // value = element index
    __syncthreads();
    const int a = addr_elems[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

static int check(const char* what, const std::vector<int>& addr) {
    int* d_a = nullptr;
    short* d_o = nullptr;
    hipMalloc(&d_a, 64 * sizeof(int));
    hipMalloc(&d_o, 256 * sizeof(short));
    hipMemcpy(d_a, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_a, d_o);
    std::vector<short> out(256);
    hipMemcpy(out.data(), d_o, 256 * sizeof(short), hipMemcpyDeviceToHost);
    hipFree(d_a);
    hipFree(d_o);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int g = l & ~15, i = l & 15;
            const int expect = addr[g + 4 * j + (i >> 2)] + (i & 3);
            if (out[l * 4 + j] != (short)expect) ++bad;
        }
    printf("%-40s %s (%d of 256 elements differ from the model)\n", what, bad ? "MODEL FAILS" : "model holds", bad);
    if (bad)
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d addr %5d -> %5d %5d %5d %5d\n", l, addr[l], out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    return bad;
}

int main() {
    std::vector<int> a(64);
    int bad = 0;
    for (int l = 0; l < 64; ++l) a[l] = 4 * l;                                   // contiguous [4 groups][4 rows][16]
    bad += check("contiguous image", a);
    for (int l = 0; l < 64; ++l) {                                               // rows anywhere: row stride 72 elements, groups far apart
        const int g = l >> 4, i = l & 15;
        a[l] = g * 1536 + (3 - (i >> 2)) * 72 + 4 * (i & 3) + 8 * (g & 1);
    }
    bad += check("free row stride / reversed rows", a);
    srand(7);
    for (int l = 0; l < 64; ++l) a[l] = 4 * (rand() % 2000);                     // any 8-byte aligned piece per lane
    bad += check("random 8-byte aligned pieces", a);
    for (int l = 0; l < 64; ++l) a[l] = 128;
    bad += check("one address for all lanes", a);
    printf(bad ? "RESULT: model does NOT describe ds_read_b64_tr_b16\n" : "RESULT: model describes ds_read_b64_tr_b16\n");
    return bad ? 1 : 0;
}
