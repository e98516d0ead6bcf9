// Stand-alone user of the C ABI (include/vlsat.h): no Python, no PyTorch -- only the HIP runtime for device memory.
// It does what a reference maintainer's binding does: create -> load weights -> finalise ->
// plan -> forward, then compares the four outputs with the expected values it was given.
//
//   c_abi_demo <dir>      <dir> holds raw little-endian files written by tests/test_hip_c_abi_demo.py:
//     meta.txt     "n_layers N E P n_weights"
//     weights.bin  per tensor: int32 name_len, name bytes, int64 count, float32[count]
//     obj_points.bin f32[N,3,P]  obj_2d_feats.bin f32[N,512]  descriptor.bin f32[N,11]
//     edges.bin i64[2,E]  batch_ids.bin i64[N]
//     expect_obj3d.bin f32[N,160]  expect_obj2d.bin  expect_rel3d.bin f32[E,26]  expect_rel2d.bin
// Prints the max-abs-error per output and exits 0 iff all are below 1e-3 (BASELINE tolerance).
//
// Build:  g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_demo.cpp \
//             -L cvpr2023-vlsat_amd -lvlsat_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/cvpr2023-vlsat_amd:/opt/rocm/lib -o examples/c_abi_demo
#include <hip/hip_runtime_api.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "vlsat.h"

#define CK(expr) do { int rc_ = (expr); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, vlsat_last_error()); return 2; } } while (0)
#define HK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); return 3; } } while (0)

template <class T> static std::vector<T> slurp(const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(4); }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T));
    if (fread(v.data(), 1, n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path.c_str()); exit(4); }
    fclose(f);
    return v;
}
template <class T> static int to_dev(const std::vector<T>& h, T** d) {
    HK(hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(T) + 16));
    HK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 1; }
    const std::string dir = std::string(argv[1]) + "/";
    int L = 0, n_w = 0; long N = 0, E = 0; int P = 0;
    {
        FILE* f = fopen((dir + "meta.txt").c_str(), "r");
        if (!f || fscanf(f, "%d %ld %ld %d %d", &L, &N, &E, &P, &n_w) != 5) { fprintf(stderr, "bad meta.txt\n"); return 1; }
        fclose(f);
    }
    printf("%s\n", vlsat_version());
    VlsatDims dims = {L, 8, 256, 0, 3, 160, 26, (float)std::log(1.0 / 0.07), 1, 1, 0};
    vlsat_handle h = nullptr;
    CK(vlsat_create(&dims, &h));                                             // Mmgnet.__init__
    {
        const std::vector<char> blob = slurp<char>(dir + "weights.bin");      // BaseModel.load
        size_t o = 0;
        for (int i = 0; i < n_w; ++i) {
            int32_t nl; memcpy(&nl, &blob[o], 4); o += 4;
            const std::string name(&blob[o], nl); o += nl;
            int64_t cnt; memcpy(&cnt, &blob[o], 8); o += 8;
            CK(vlsat_load_weight(h, name.c_str(), reinterpret_cast<const float*>(&blob[o]), (size_t)cnt));
            o += (size_t)cnt * 4;
        }
    }
    CK(vlsat_finalize_weights(h));
    const auto edges = slurp<int64_t>(dir + "edges.bin"), bid = slurp<int64_t>(dir + "batch_ids.bin");
    vlsat_plan plan = nullptr;
    CK(vlsat_plan_create(h, bid.data(), edges.data(), N, E, P, &plan));       // MMG.forward's graph bookkeeping
    float *pts, *f2d, *desc, *o3, *o2, *r3, *r2;
    if (to_dev(slurp<float>(dir + "obj_points.bin"), &pts) || to_dev(slurp<float>(dir + "obj_2d_feats.bin"), &f2d) ||
        to_dev(slurp<float>(dir + "descriptor.bin"), &desc)) return 3;
    HK(hipMalloc(reinterpret_cast<void**>(&o3), N * 160 * 4)); HK(hipMalloc(reinterpret_cast<void**>(&o2), N * 160 * 4));
    HK(hipMalloc(reinterpret_cast<void**>(&r3), (E + 1) * 26 * 4)); HK(hipMalloc(reinterpret_cast<void**>(&r2), (E + 1) * 26 * 4));
    hipStream_t s;
    HK(hipStreamCreate(&s));
    CK(vlsat_forward(h, plan, pts, f2d, desc, o3, o2, r3, r2, s));            // Mmgnet.forward(istrain=False)
    HK(hipStreamSynchronize(s));
    int bad = 0;
    const struct { const char* name; float* dev; long count; } outs[4] = {
        {"obj3d", o3, N * 160}, {"obj2d", o2, N * 160}, {"rel3d", r3, E * 26}, {"rel2d", r2, E * 26}};
    for (const auto& o : outs) {
        std::vector<float> got(o.count);
        HK(hipMemcpy(got.data(), o.dev, o.count * 4, hipMemcpyDeviceToHost));
        const auto want = slurp<float>(dir + "expect_" + o.name + ".bin");
        double err = 0;
        for (long i = 0; i < o.count; ++i) err = std::fmax(err, std::fabs((double)got[i] - want[i]));
        printf("%-6s max-abs-err %.3e over %ld values\n", o.name, err, o.count);
        if (!(err < 1e-3) || (long)want.size() != o.count) bad = 1;
    }
    vlsat_plan_destroy(plan);
    vlsat_destroy(h);
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
