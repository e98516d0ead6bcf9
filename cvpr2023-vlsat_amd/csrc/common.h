// Shared host/device helpers for libvlsat_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Release and experiment builds.  `python cvpr2023-vlsat_amd/build.py` builds the RELEASE library: no timing-ablation kernels, no
// garbage-result switches, only the debug options the tests use.  `build.py --experiments` adds -DVLSAT_EXPERIMENTS and writes
// tools/bin/libvlsat_hip_exp.so (loaded by `bench.py --lib` and the probes under tools/): ablation instantiations of the GEMM
// kernels, FlashSplit::ablate / GemmArgs::ablate honoured, every experiment switch of vlsat_debug_option accepted.
#ifdef VLSAT_EXPERIMENTS
constexpr bool kExperiments = true;
#else
constexpr bool kExperiments = false;
#endif

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

// Reductions across the two 32-lane halves of a wave with the gfx950 register swap (v_permlane32_swap_b32 a, b exchanges lanes
// 32..63 of a with lanes 0..31 of b: from a = b = x every lane of a then holds the LOWER half's value of its column and every lane
// of b the UPPER half's) instead of __shfl_xor(x, 32), which hipcc turns into a ds_bpermute_b32: an LDS-queue instruction with an
// address register and an lgkmcnt wait behind whatever fragment reads are in flight.  max and + are commutative, so the results
// equal fmaxf(x, __shfl_xor(x, 32)) / x + __shfl_xor(x, 32) bit for bit in both halves.
__device__ __forceinline__ void half_swap(float x, float& lo, float& hi) {
    typedef unsigned u32x2_hs __attribute__((ext_vector_type(2)));
    const u32x2_hs r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r[0]);
    hi = __uint_as_float(r[1]);
}
__device__ __forceinline__ float half_max(float x) { float a, b; half_swap(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float x) { float a, b; half_swap(x, a, b); return a + b; }

// XCD-aware bijective block remap: hardware round-robins consecutive block ids over the 8
// XCDs; this gives every XCD a contiguous range of virtual ids so neighbouring tiles (which
// share an operand panel) hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Split-pair format of the bf16 modes (one 32-bit word per element, same footprint and addressing as fp32):
// upper half = bf16 hi = rne(x), lower half = bf16 lo = rne(x - hi); x ~= hi + lo to 2^-17 relative.
__device__ __forceinline__ float pack_split(float x) {
    const __bf16 h = (__bf16)x;                                        // v_cvt_pk_bf16_f32 (round to nearest even)
    const __bf16 l = (__bf16)(x - (float)h);
    const unsigned hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l);
    return __uint_as_float((hb << 16) | lb);
}
__device__ __forceinline__ float unpack_split(float w) {
    const unsigned u = __float_as_uint(w);
    return __uint_as_float(u & 0xffff0000u) + __uint_as_float(u << 16);
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
