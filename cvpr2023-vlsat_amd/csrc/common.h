// Shared host/device helpers for libvlsat_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error plumbing (host) -------------------------------------------------------------
namespace vlsat {
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
}  // namespace vlsat

#define VLSAT_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return vlsat::fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));     \
    } while (0)

#define VLSAT_LAUNCH_CHECK(what)                                                           \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess)                                                              \
            return vlsat::fail(-2, std::string("launch ") + what + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- device helpers ----------------------------------------------------------------------
#if defined(__HIPCC__)

// Row of element r (0..15) of a 32x32 MFMA C/D fragment held by a lane of half hi = lane>>5.
// Column is lane & 31.  (cdna_hip_programming.md §3: row = (r&3) + 8*(r>>2) + 4*hi.)
__device__ __forceinline__ int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// XCD-aware bijective block remap: hardware round-robins consecutive block ids over the 8
// XCDs; this gives every XCD a contiguous range of virtual ids so neighbouring tiles (which
// share an operand panel) hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
#endif
