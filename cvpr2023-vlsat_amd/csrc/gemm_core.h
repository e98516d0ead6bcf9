// fp32 MFMA building blocks shared by the GEMM, PointNet and gate kernels (gfx950).
//
// Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD).  Operand layout
// (cdna_hip_programming.md §3): lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// the 32x32 result has column l&31 and rows crow32(r, l>>5), r = 0..15.
//
// K-permutation trick: a GEMM sums over k, so which k a (step, lane-half) pair covers is free
// as long as A and B agree.  Lane (row, hi) reads FOUR consecutive k (one ds_read_b128) at
// k = kg*8 + 4*hi .. +3 and feeds them to four consecutive MFMAs; step s of k-group kg then
// covers k = {kg*8 + s, kg*8 + 4 + s}.  This keeps both operand tiles in their natural
// row-major [row][k] layout (same as HBM), so staging is a straight float4 copy.
//
// LDS tiles are [rows][BK=32] with a row pitch of 36 floats: the 16-byte pad makes the
// ds_read_b128 of 16 different rows (one lane group) hit 16 disjoint 4-bank groups
// (start bank = 36*row mod 64 -> conflict-free, MI355X_MICROARCH.md §LDS).
#pragma once
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace vlsat {

constexpr int BK = 32;    // k-slice held in LDS per pipeline stage

// residual element (ml, nl) of the tile whose first element is rbase = resid + m0 * ldr + n0, in the operand's format:
// 0 fp32, 1 split-pair word, 2 half row (bf16 at byte 2 * column of the fp32-pitched row)
// ---- accumulator-side access of the GEMM kernels -------------------------------------------------------------------------
// The GEMM pipes issue the transposed product (mma_slice SWAP): element r of lane (li = lane & 31, hi = lane >> 5) in the
// 32x32 tile (tm, tn) of wave (wm, wn) is C[(wm*TM + tm)*32 + li][(wn*TN + tn)*32 + crow32(r, hi)] -- one row per lane,
// columns in four runs of 4 (r = 4g .. 4g+3 -> columns 8g + 4hi .. +3).  Everything below therefore moves 16 bytes per lane
// and instruction (8 for bf16 half rows) where the untransposed layout needed four scalar accesses.
template <int FMT>
__device__ __forceinline__ float load_resid(const float* rbase, int ml, int nl, int ldr, int n0) {
    if (FMT == 2) {
        const unsigned short* row = reinterpret_cast<const unsigned short*>(rbase - n0 + (size_t)ml * ldr);
        return __uint_as_float((unsigned)row[n0 + nl] << 16);
    }
    const float rv = rbase[(unsigned)(ml * ldr + nl)];
    return FMT == 1 ? unpack_split(rv) : rv;
}
// four consecutive elements starting at column `col` (multiple of 4) of the fp32-pitched row `row`, in storage format FMT
// (0 fp32, 1 split-pair words, 2 half row: bf16 at byte 2 * column)
template <int FMT>
__device__ __forceinline__ f32x4 load_row4(const float* row, int col) {
    if (FMT == 2) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(row) + col);
        return f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(row + col);
    if (FMT == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = unpack_split(v[c]);
    }
    return v;
}
// The storage format is a launch constant, but a per-element branch on it keeps hipcc from batching a lane's loads
// (measured: out-proj + residual 97 -> 79 TFLOP/s in fp32, 445 -> 161 in bf16): callers hoist it with this dispatcher.
template <class F>
__device__ __forceinline__ void with_format(int fmt, F&& f) {
    if (fmt == 2) f(std::integral_constant<int, 2>{});
    else if (fmt == 1) f(std::integral_constant<int, 1>{});
    else f(std::integral_constant<int, 0>{});
}
__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Accumulator init of a tile = its additive operands: resid_scale * resid + g0[gi0[row]] + g1[gi1[row]] (ADD bits 0 / 1 / 2;
// compile time, so the loads of a lane are branch-free and batched).  The first MFMA of the tile takes them as C-in.
// PLAIN (the exact-fp32 kernels): operands and output are fp32 and c_scale is 1 -- the format dispatch folds away.
template <int TM, int TN, int ADD, bool PLAIN = false>
__device__ __forceinline__ void tile_init(const GemmArgs& p, int m0, int n0, int wm, int wn, int lane, f32x16 (&acc)[TM][TN]) {
    // 32-bit offsets from wave-uniform bases; the asm makes the lane id and pitches opaque per tile (otherwise LICM hoists
    // all the per-lane offsets out of a persistent kernel's tile loop and it spills)
    int ldr = p.ldr, ldg0 = p.ldg0, ldg1 = p.ldg1, lv = lane;
    asm volatile("" : "+s"(ldr), "+s"(ldg0), "+s"(ldg1), "+v"(lv));
    const int li = lv & 31, hi = lv >> 5;
    const bool vec = n0 + (wn * TN + TN) * 32 <= p.N && ((ldr | ldg0 | ldg1) & 3) == 0 &&
                     (!(ADD & 1) || aligned16(p.resid)) && (!(ADD & 2) || aligned16(p.g0)) && (!(ADD & 4) || aligned16(p.g1));
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    if (!(ADD & 1) && !PLAIN && p.g_f16) {       // fp16 half-row tables (GemmArgs::g_f16; the launcher admits them for N % 256 == 0 only: every tile's columns are inside N): 8 bytes per lane and load
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            int m = m0 + (wm * TM + tm) * 32 + li;
            m = m < p.M ? m : p.M - 1;
            const _Float16* g0row = (ADD & 2) ? reinterpret_cast<const _Float16*>(p.g0 + (size_t)p.gi0[m] * ldg0) : nullptr;
            const _Float16* g1row = (ADD & 4) ? reinterpret_cast<const _Float16*>(p.g1 + (size_t)p.gi1[m] * ldg1) : nullptr;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + (wn * TN + tn) * 32 + 8 * g + 4 * hi;
                    f32x4 x = {0.f, 0.f, 0.f, 0.f};
                    if (ADD & 2) x = __builtin_convertvector(*reinterpret_cast<const f16x4*>(g0row + col), f32x4);
                    if (ADD & 4) x += __builtin_convertvector(*reinterpret_cast<const f16x4*>(g1row + col), f32x4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[tm][tn][4 * g + c] = x[c];
                }
        }
        return;
    }
    if (vec) {
        with_format((ADD & 1) && !PLAIN ? p.r_split : 0, [&](auto fmt) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                int m = m0 + (wm * TM + tm) * 32 + li;
                m = m < p.M ? m : p.M - 1;
                const float* rrow = (ADD & 1) ? p.resid + (size_t)m * ldr : nullptr;
                const float* g0row = (ADD & 2) ? p.g0 + (size_t)p.gi0[m] * ldg0 : nullptr;
                const float* g1row = (ADD & 4) ? p.g1 + (size_t)p.gi1[m] * ldg1 : nullptr;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = n0 + (wn * TN + tn) * 32 + 8 * g + 4 * hi;
                        f32x4 x = {0.f, 0.f, 0.f, 0.f};
                        if (ADD & 1) x = load_row4<decltype(fmt)::value>(rrow, col) * p.resid_scale;
                        if (ADD & 2) x += *reinterpret_cast<const f32x4*>(g0row + col);
                        if (ADD & 4) x += *reinterpret_cast<const f32x4*>(g1row + col);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[tm][tn][4 * g + c] = x[c];
                    }
            }
        });
        return;
    }
    // tiles cut by N (or operands that are not 16-byte addressable): element by element, clamped
    with_format((ADD & 1) && !PLAIN ? p.r_split : 0, [&](auto fmt) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            int m = m0 + (wm * TM + tm) * 32 + li;
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int n = n0 + (wn * TN + tn) * 32 + crow32(r, hi);
                    n = n < p.N ? n : p.N - 1;
                    float x = 0.f;
                    if (ADD & 1) x = p.resid_scale * load_resid<decltype(fmt)::value>(p.resid + (size_t)m * ldr, 0, n, 0, 0);
                    if (ADD & 2) x += p.g0[(size_t)p.gi0[m] * ldg0 + n];
                    if (ADD & 4) x += p.g1[(size_t)p.gi1[m] * ldg1 + n];
                    acc[tm][tn][r] = x;
                }
        }
    });
}

// Epilogue of a finished tile: row scale, bias, activation, final scale, store in the output format (c_split).  Everything
// is straight-line code under WAVE-UNIFORM branches (per-element branches on the runtime flags cost ~30 % of the tile
// time in the first version of the kernel).  BM / BN: the block tile (for the interior test).
template <int TM, int TN, bool PLAIN = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& p, int m0, int n0, int BM, int BN, int wm, int wn, int lane,
                                              f32x16 (&acc)[TM][TN]) {
    const bool c_f16 = !PLAIN && p.c_f16_cols > 0 && (n0 + BN <= p.c_f16_cols || p.c_f16_cols >= p.N);      // (c_f16_cols == N: the whole output, whatever N is)      // this block tile's columns are fp16 half rows (GemmArgs::c_f16_cols)
    const int c_split = PLAIN ? 0 : p.c_split;
    int ldc = p.ldc, lv = lane;
    asm volatile("" : "+s"(ldc), "+v"(lv));                          // see tile_init: no LICM of the store offsets
    const int li = lv & 31, hi = lv >> 5;
    const bool cols_in = n0 + (wn * TN + TN) * 32 <= p.N;            // this wave's columns are all inside N
    if (p.rowscale) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            int m = m0 + (wm * TM + tm) * 32 + li;
            m = m < p.M ? m : p.M - 1;
            const float rs = p.rowscale[m];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= rs;
        }
    }
    if (p.bias) {
        if (cols_in && aligned16(p.bias)) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n0 + (wn * TN + tn) * 32 + 8 * g + 4 * hi);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[tm][tn][4 * g + c] += b[c];
                }
        } else {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int n = n0 + (wn * TN + tn) * 32 + crow32(r, hi);
                    n = n < p.N ? n : p.N - 1;
                    const float b = p.bias[n];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn][r] += b;
                }
        }
    }
    if (p.act == ACT_RELU) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = fmaxf(acc[tm][tn][r], 0.f);
    } else if (p.act == ACT_SIGMOID) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 1.f / (1.f + __expf(-acc[tm][tn][r]));
    }
    if (!PLAIN && p.c_scale != 1.f) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] *= p.c_scale;
    }
    const bool interior = m0 + BM <= p.M && cols_in && (ldc & 3) == 0 && aligned16(p.C);
    if (c_f16) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = m0 + (wm * TM + tm) * 32 + li;
            _Float16* crow = reinterpret_cast<_Float16*>(p.C + (size_t)(m < p.M ? m : 0) * ldc);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + (wn * TN + tn) * 32 + 8 * g + 4 * hi;
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fminf(fmaxf(acc[tm][tn][4 * g + c], -65504.f), 65504.f);
                    const f16x4 hv = __builtin_convertvector(v, f16x4);
                    if (interior) *reinterpret_cast<f16x4*>(crow + col) = hv;
                    else if (m < p.M) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (col + c < p.N) crow[col + c] = hv[c];
                    }
                }
        }
        return;
    }
    if (interior) {
        with_format(c_split, [&](auto fmt) {
            constexpr int FMT = decltype(fmt)::value;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                float* crow = p.C + (size_t)(m0 + (wm * TM + tm) * 32 + li) * ldc;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = n0 + (wn * TN + tn) * 32 + 8 * g + 4 * hi;
                        f32x4 v;
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = acc[tm][tn][4 * g + c];
                        if (FMT == 2) {               // half rows: four bf16 = one 8-byte store
                            uint2 w;
                            w.x = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16);
                            w.y = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16);
                            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(crow) + col) = w;
                        } else {
                            if (FMT == 1) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) v[c] = pack_split(v[c]);
                            }
                            *reinterpret_cast<f32x4*>(crow + col) = v;
                        }
                    }
            }
        });
        return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + (wm * TM + tm) * 32 + li;
        float* crow = p.C + (size_t)(m < p.M ? m : 0) * ldc;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * TN + tn) * 32 + crow32(r, hi);
                if (m < p.M && n < p.N) {
                    const float v = acc[tm][tn][r];
                    if (c_split == 2) reinterpret_cast<unsigned short*>(crow)[n] = __builtin_bit_cast(unsigned short, (__bf16)v);
                    else crow[n] = c_split ? pack_split(v) : v;
                }
            }
    }
}
constexpr int LDT = 36;   // LDS row pitch in floats (BK + 4 pad)

// Issue the global loads of a [ROWS][BK] slice (rows row0.., k0..k0+31) into registers.
// 256 threads, ROWS/32 float4 each; 8 consecutive threads cover one 128-byte row segment.
// Rows beyond row_last are clamped (their results are discarded by the caller).
template <int ROWS>
__device__ __forceinline__ void stage_load(const float* __restrict__ base, int ld, int row0, int row_last,
                                           int k0, f32x4 (&regs)[ROWS / 32], int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        int row = row0 + (idx >> 3);
        row = row < row_last ? row : row_last;
        regs[i] = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + k0 + (idx & 7) * 4);
    }
}

template <int ROWS>
__device__ __forceinline__ void stage_relu(f32x4 (&regs)[ROWS / 32]) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) regs[i][c] = fmaxf(regs[i][c], 0.f);
}

template <int ROWS>
__device__ __forceinline__ void stage_store(float* __restrict__ lds, const f32x4 (&regs)[ROWS / 32], int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        *reinterpret_cast<f32x4*>(lds + (idx >> 3) * LDT + (idx & 7) * 4) = regs[i];
    }
}

// One BK=32 slice of C[TM*32][TN*32] += A . B^T for one wave.
// sA: LDS row 0 of this wave's A rows ([TM*32][LDT]); sB likewise for its B rows.
// SWAP = false: lane holds C column (lane & 31) and 16 rows crow32(r, lane >> 5) of each 32x32 tile (PointNet kernel);
// SWAP = true: the transposed product B.A^T -- lane holds C ROW (lane & 31) and 16 COLUMNS crow32(r, lane >> 5), i.e. four
// runs of 4 consecutive columns, so every accumulator-side access of the GEMM kernels (residual, gathered rows, bias,
// stores) is a 16-byte one (tile_init / tile_epilogue below).
template <int TM, int TN, int PITCH_A = LDT, int PITCH_B = LDT, bool SWAP = false>
__device__ __forceinline__ void mma_slice(const float* __restrict__ sA, const float* __restrict__ sB,
                                          f32x16 (&acc)[TM][TN], int lane) {
    const int li = lane & 31, hi = lane >> 5;
    // Two fragment sets in flight: the ds_reads of k-group kg+2 are issued right after the MFMAs
    // of group kg have ISSUED (they read their operands at issue) and are consumed one whole
    // group (TM*TN*4 MFMAs) later, so LDS latency never reaches the matrix pipe.  The
    // sched_barriers pin that order (left alone, hipcc re-uses one register set and places the
    // reads one MFMA ahead of their use).
    f32x4 a[2][TM], b[2][TN];
    auto load = [&](int set, int kg) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            a[set][tm] = *reinterpret_cast<const f32x4*>(sA + (tm * 32 + li) * PITCH_A + kg * 8 + hi * 4);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            b[set][tn] = *reinterpret_cast<const f32x4*>(sB + (tn * 32 + li) * PITCH_B + kg * 8 + hi * 4);
    };
    load(0, 0);
    load(1, 1);
#pragma unroll
    for (int kg = 0; kg < BK / 8; ++kg) {
        const int set = kg & 1;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[set][tn][s], a[set][tm][s], acc[tm][tn], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][tm][s], b[set][tn][s], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kg + 2 < BK / 8) {
            load(set, kg + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// Operand pipelines of the persistent GEMM (gemm_f32.hip): how a [BM|BN][32] k-slice travels
// HBM -> registers -> LDS -> MFMA fragments.  PipeF32 is the exact-fp32 path above.
struct NoCtx {
    template <class Args> __device__ __forceinline__ NoCtx(const Args&, int) {}
};

template <int BM, int BN>
struct PipeF32 {
    static constexpr bool PREFETCH = false;
    static constexpr int TM = BM / 64, TN = BN / 64;
    static constexpr int STAGE_BYTES = (BM + BN) * LDT * 4;
    struct Regs { f32x4 a[BM / 32], b[BN / 32]; };
    using Ctx = NoCtx;
    template <class Args>
    static __device__ __forceinline__ void load(const Ctx&, const Args& p, int m0, int n0, int k0, Regs& r, int tid, char*) {
        stage_load<BM>(p.A, p.lda, m0, p.M - 1, k0, r.a, tid);
        stage_load<BN>(p.W, p.ldw, n0, p.N - 1, k0, r.b, tid);
    }
    static __device__ __forceinline__ void store(char* stage, Regs& r, int tid, int relu_a) {
        float* s = reinterpret_cast<float*>(stage);
        if (relu_a) stage_relu<BM>(r.a);
        stage_store<BM>(s, r.a, tid);
        stage_store<BN>(s + BM * LDT, r.b, tid);
    }
    static __device__ __forceinline__ void mma(const char* stage, int wm, int wn, f32x16 (&acc)[TM][TN], int lane, int = 0) {
        const float* s = reinterpret_cast<const float*>(stage);
        mma_slice<TM, TN, LDT, LDT, true>(s + (wm * TM * 32) * LDT, s + BM * LDT + (wn * TN * 32) * LDT, acc, lane);
    }
};

// Exact-fp32 path with LDS-direct staging: `buffer_load_dwordx4 ... lds` moves the k-slice HBM/L2 -> LDS
// without the VGPR round trip and without ds_write.  tools/vmem_issue_probe.hip: a global_load_dwordx4 plus
// its ds_write_b128 takes ~64 matrix-pipe cycles away from the issuing SIMD, a buffer_load ... lds ~40; at 8
// loads per wave per 64-MFMA slice that is the difference between 12.5 % and 8 % of the slice.
//
// The LDS-direct write puts lane l's 16 bytes at base + 16 l (tools/lds_dma_check.hip), so one wave
// instruction fills 8 consecutive UNPADDED 128-byte rows and padding cannot be used against bank conflicts.
// Instead the 16-byte chunks of a row are XOR-swizzled: logical chunk c of row r lives at chunk
// c ^ ((r >> 1) & 7).  A ds_read_b128 by 16 consecutive lanes (rows r..r+15, same logical chunk) then
// touches 2 row parities x 8 distinct chunks = all 64 banks exactly once.  The swizzle is applied on the
// global side (the lane that writes physical chunk j of a row fetches logical chunk j ^ sw from HBM -- the 8
// lanes of a row still cover one full 128-byte line) and on the fragment reads.
// Rows past the matrix edge are not clamped: the buffer descriptor's num_records makes them read as zero
// (the whole offset travels in the VGPR: on gfx9 the range check does not see the SGPR offset).
template <int BM, int BN>
struct PipeF32Dma {
    static constexpr bool PREFETCH = false;          // (the A-panel touch of PipeSplitDma measured -2 % here: 4096-cycle slices already cover the load)
    static constexpr int TM = BM / 64, TN = BN / 64;
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 4;
    struct Regs {};
    struct Ctx {
        int na, nw;                         // bytes addressable behind A / W (buffer num_records)
        unsigned va, vw;                    // per-lane byte offsets inside an 8-row group (swizzled chunk)
        template <class Args>
        __device__ __forceinline__ Ctx(const Args& p, int tid) {
            const int wave = tid >> 6, l = tid & 63;
            na = (int)(((size_t)(p.M - 1) * p.lda + p.K) * 4);
            nw = (int)(((size_t)(p.N - 1) * p.ldw + p.K) * 4);
            const int row = 8 * wave + (l >> 3);                       // row inside a 32-row instruction group
            const int chunk = (l & 7) ^ ((row >> 1) & 7);              // logical chunk this lane fetches
            va = (unsigned)(row * p.lda + 4 * chunk) * 4u;
            vw = (unsigned)(row * p.ldw + 4 * chunk) * 4u;
        }
    };
    // instruction i of a wave covers tile rows 32 i + 8 wave .. + 7 (so the swizzle term does not depend on i)
    static __device__ __forceinline__ void load(const Ctx& c, const GemmArgs& p, int m0, int n0, int k0, Regs&, int tid, char* stage) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        float* s = reinterpret_cast<float*>(stage) + wave * 8 * BK;
        // (the descriptor type cannot be a struct member: the host pass of hipcc rejects it; these are 8 SGPRs
        //  of loop-invariant scalar code)
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, c.na, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, c.nw, 0x00020000);
#pragma unroll
        for (int i = 0; i < BM / 32; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, s + i * 32 * BK, 16, c.va + (unsigned)(((m0 + 32 * i) * p.lda + k0) * 4), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, s + (BM + i * 32) * BK, 16, c.vw + (unsigned)(((n0 + 32 * i) * p.ldw + k0) * 4), 0, 0, 0);
    }
    // the slice has had a whole slice of MFMAs to land; the caller's barrier publishes it
    static __device__ __forceinline__ void store(char*, Regs&, int, int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    static __device__ __forceinline__ void store_keep1() { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
    static __device__ __forceinline__ void mma(const char* stage, int wm, int wn, f32x16 (&acc)[TM][TN], int lane, int = 0) {
        const int li = lane & 31, hi = lane >> 5;
        const int sw = (li >> 1) & 7;
        const float* sA = reinterpret_cast<const float*>(stage) + (wm * TM * 32 + li) * BK + 4 * ((hi ^ sw) & 1);
        const float* sB = sA + (BM + wn * TN * 32 - wm * TM * 32) * BK;
        const int y = sw >> 1;
        f32x4 a[2][TM], b[2][TN];
        auto load = [&](int set, int kg) {
            const int o = 8 * (kg ^ y);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[set][tm] = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + o);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[set][tn] = *reinterpret_cast<const f32x4*>(sB + tn * 32 * BK + o);
        };
        load(0, 0);
        load(1, 1);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            const int set = kg & 1;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[set][tn][s], a[set][tm][s], acc[tm][tn], 0, 0, 0);   // transposed: see mma_slice
            __builtin_amdgcn_sched_barrier(0);
            if (kg + 2 < BK / 8) {
                load(set, kg + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
};

// Split-bf16 path (BASELINE configs[2], "bf16 MFMA for the GEMMs"): v_mfma_f32_32x32x16_bf16 with
// fp32 accumulate.  TERMS = 1: operands rounded to bf16 (one MFMA per k-step; 2^-9 relative
// input error -> ~2e-2 on the x14.29 object logits, misses the 1e-2 tolerance).
// TERMS = 3: a = a_hi + a_lo, w = w_hi + w_lo with both parts bf16;
//   a.w ~= a_hi.w_hi + a_lo.w_hi + a_hi.w_lo   (dropped a_lo.w_lo ~ 2^-16 relative),
// three MFMAs at 16x the fp32-MFMA rate = 5.3x the fp32 matrix throughput at ~1e-5 error.
// A stays fp32 in HBM and is split while staging; weights are pre-split once (Whi/Wlo, [N,K] bf16).
// LDS: bf16 planes [rows][32] with an 80-byte row pitch (conflict-free ds_read_b128 of 8 k).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// 16-bit operand MFMA on registers declared as bf16x8: F16 reinterprets the same bits as fp16 (GemmArgs::half_f16)
template <bool F16>
__device__ __forceinline__ f32x16 mfma_h(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int BM, int BN, int TERMS>
struct PipeBF16 {
    static constexpr bool PREFETCH = false;
    static constexpr int TM = BM / 64, TN = BN / 64;
    static constexpr int PH = 40;                                  // row pitch in bf16 elements (80 B)
    static constexpr int PL = TERMS == 1 ? 1 : 2;                  // planes: hi (, lo)
    static constexpr int A_PLANE = BM * PH * 2, W_PLANE = BN * PH * 2;       // bytes
    static constexpr int STAGE_BYTES = (A_PLANE + W_PLANE) * PL;
    struct Regs { f32x4 a[BM / 32]; uint4 w[PL][BN / 64]; };
    using Ctx = NoCtx;
    template <class Args>
    static __device__ __forceinline__ void load(const Ctx&, const Args& p, int m0, int n0, int k0, Regs& r, int tid, char*) {
        stage_load<BM>(p.A, p.lda, m0, p.M - 1, k0, r.a, tid);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            const uint16_t* w = pl ? p.Wlo : p.Whi;
#pragma unroll
            for (int i = 0; i < BN / 64; ++i) {
                const int idx = tid + 256 * i;
                int row = n0 + (idx >> 2);
                row = row < p.N ? row : p.N - 1;
                r.w[pl][i] = *reinterpret_cast<const uint4*>(w + (size_t)row * p.ldw + k0 + (idx & 3) * 8);
            }
        }
    }
    static __device__ __forceinline__ void store(char* stage, Regs& r, int tid, int relu_a) {
        if (relu_a) stage_relu<BM>(r.a);
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int idx = tid + 256 * i;
            const int off = ((idx >> 3) * PH + (idx & 7) * 4) * 2;
            const bf16x4 h = __builtin_convertvector(r.a[i], bf16x4);
            *reinterpret_cast<bf16x4*>(stage + off) = h;
            if (PL == 2) {
                const f32x4 rest = r.a[i] - __builtin_convertvector(h, f32x4);
                *reinterpret_cast<bf16x4*>(stage + A_PLANE + off) = __builtin_convertvector(rest, bf16x4);
            }
        }
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
#pragma unroll
            for (int i = 0; i < BN / 64; ++i) {
                const int idx = tid + 256 * i;
                *reinterpret_cast<uint4*>(stage + A_PLANE * PL + W_PLANE * pl + ((idx >> 2) * PH + (idx & 3) * 8) * 2) = r.w[pl][i];
            }
    }
    static __device__ __forceinline__ void mma(const char* stage, int wm, int wn, f32x16 (&acc)[TM][TN], int lane, int = 0) {
        const int li = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[PL][TM], w[PL][TN];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    a[pl][tm] = *reinterpret_cast<const bf16x8*>(stage + A_PLANE * pl + (((wm * TM + tm) * 32 + li) * PH + ks * 16 + hi * 8) * 2);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    w[pl][tn] = *reinterpret_cast<const bf16x8*>(stage + A_PLANE * PL + W_PLANE * pl + (((wn * TN + tn) * 32 + li) * PH + ks * 16 + hi * 8) * 2);
            }
            // term-major order (small terms first): the TM x TN accumulators take turns, so no MFMA waits for the one before it
            if (PL == 2) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0][tn], a[1][tm], acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PL - 1][tn], a[0][tm], acc[tm][tn], 0, 0, 0);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0][tn], a[0][tm], acc[tm][tn], 0, 0, 0);
        }
    }
};


// bf16 / split-bf16 path with LDS-direct staging (round 2; what the forward uses in the bf16 modes).
// A stays fp32 in HBM and travels HBM/L2 -> LDS exactly like PipeF32Dma's (same swizzled 128-byte rows); it is split
// into bf16 hi/lo on the FRAGMENT READ side: 8 consecutive k = two ds_read_b128, 24 VALU (cvt_pk, shift, sub, cvt_pk),
// hidden behind the 32-cycle bf16 MFMAs.  The pre-split weight planes (Whi / Wlo, [N,K] bf16) are DMA'd as 64-byte
// rows, 16 rows per wave instruction (lane l -> row l>>2, 16-byte chunk l&3), chunks XOR-swizzled with (row>>2)&3 so
// that the ds_read_b128 lane groups of MI355X_MICROARCH.md ({0-3,12-15,20-27} ...) touch every bank once.
// Against PipeBF16 (VGPR staging, split while storing to LDS) this removes the global_load -> VGPR -> ds_write round
// trip (8 loads + 12 ds_writes per wave per slice, ~64 matrix-pipe cycles each, tools/vmem_issue_probe.hip).
// k-slot convention (A and W agree): kstep ks, lane half hi, element e covers k = 16 ks + 8 hi + e.
// AFMT: how A is stored.  0 = fp32.  1 = SPLIT-PAIR words, the format the split-bf16 mode keeps its edge tensors in (one
// 32-bit word per element: bf16 hi = rne(x) in the upper half, bf16 lo = rne(x - hi) in the lower half; written by the
// producing kernel's epilogue, common.h pack_split): same footprint and addressing as fp32, so the DMA staging is
// unchanged, and the fragment-side split shrinks from ~24 VALU per 8 elements (cvt_pk, shift, sub, cvt_pk) to 8
// v_perm_b32.  2 = HALF rows, the format of the single-rounding modes: the row keeps its fp32 pitch but only its first
// K * 2 bytes are used, K bf16 values -- half the HBM traffic; staged as 64-byte rows exactly like the weight planes,
// fragments are read with one ds_read_b128 and need no VALU at all (TERMS = 1 only).
template <int BM, int BN, int TERMS, int AFMT = 0>
struct PipeSplitDma {
    static constexpr bool AS = AFMT == 1;
    static constexpr bool AH = AFMT >= 2;             // (3: the half rows and the weight plane hold fp16, products on the f16 MFMA)
    static constexpr bool F16 = AFMT == 3;
    static_assert(!AH || TERMS == 1, "half-row operands carry no low part");
    // With bf16 MFMAs a k-slice is 256 (bf16) to 768 (bf16x3) matrix-pipe cycles per wave, far less than the latency of
    // an A line that comes from HBM / the Infinity Cache, and LDS cannot hold enough slices in flight to cover it: the
    // kernel touches the A lines of the slice `prefetch` steps ahead (one dword per line into a dead register) so
    // that the LDS-direct loads that follow hit the XCD's L2.  See gemm_f32.hip.
    static constexpr bool PREFETCH = true;
    static constexpr int TM = BM / 64, TN = BN / 64;
    static constexpr int PL = TERMS == 1 ? 1 : 2;
    static constexpr int A_BYTES = AH ? BM * BK * 2 : BM * BK * 4, W_PLANE = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + W_PLANE * PL;
    struct Regs {};
    struct Ctx {
        int na, nw;                         // bytes addressable behind A / a W plane
        unsigned va, vw;                    // per-lane byte offsets inside an instruction's row group (swizzled chunk)
        template <class Args>
        __device__ __forceinline__ Ctx(const Args& p, int tid) {
            const int wave = tid >> 6, l = tid & 63;
            nw = (int)(((size_t)(p.N - 1) * p.ldw + p.K) * 2);
            const int wrow = 16 * wave + (l >> 2);                       // (wrow >> 2) & 3 == (l >> 4) & 3
            vw = (unsigned)(wrow * p.ldw + 8 * ((l & 3) ^ ((l >> 4) & 3))) * 2u;
            if (AH) {                                                    // 64-byte row pieces at the fp32 row pitch
                na = (int)((size_t)(p.M - 1) * p.lda * 4 + (size_t)p.K * 2);
                va = (unsigned)(wrow * p.lda * 4 + 16 * ((l & 3) ^ ((l >> 4) & 3)));
            } else {
                na = (int)(((size_t)(p.M - 1) * p.lda + p.K) * 4);
                const int row = 8 * wave + (l >> 3);
                va = (unsigned)(row * p.lda + 4 * ((l & 7) ^ ((row >> 1) & 7))) * 4u;
            }
        }
    };
    static __device__ __forceinline__ void load(const Ctx& c, const GemmArgs& p, int m0, int n0, int k0, Regs&, int tid, char* stage) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, c.na, 0x00020000);
        if (AH) {
            char* sa = stage + wave * 16 * BK * 2;
#pragma unroll
            for (int i = 0; i < BM / 64; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 64 * BK * 2, 16, c.va + (unsigned)((m0 + 64 * i) * p.lda * 4 + k0 * 2), 0, 0, 0);
        } else {
            float* sa = reinterpret_cast<float*>(stage) + wave * 8 * BK;
#pragma unroll
            for (int i = 0; i < BM / 32; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, sa + i * 32 * BK, 16, c.va + (unsigned)(((m0 + 32 * i) * p.lda + k0) * 4), 0, 0, 0);
        }
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(pl ? p.Wlo : p.Whi), 0, c.nw, 0x00020000);
            char* sw = stage + A_BYTES + pl * W_PLANE + wave * 16 * BK * 2;
#pragma unroll
            for (int i = 0; i < BN / 64; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, sw + i * 64 * BK * 2, 16, c.vw + (unsigned)(((n0 + 64 * i) * p.ldw + k0) * 2), 0, 0, 0);
        }
    }
    static __device__ __forceinline__ void store(char*, Regs&, int, int) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // the slice's loads have landed; one younger load (the prefetch touch) may stay in flight
    static __device__ __forceinline__ void store_keep1() { asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
    // half rows: the eight bf16 of a k-slot are one 16-byte LDS read; ReLU = max with +0 on the sign-magnitude patterns
    template <bool RELU>
    static __device__ __forceinline__ bf16x8 frag_half(const char* p) {
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v = *reinterpret_cast<const s16x8*>(p);
        if (RELU) v = __builtin_elementwise_max(v, s16x8{0, 0, 0, 0, 0, 0, 0, 0});
        return __builtin_bit_cast(bf16x8, v);
    }
    // 8 consecutive k of one row (two 16-byte chunks) -> the bf16 hi / lo operand registers
    template <bool RELU>
    static __device__ __forceinline__ void split8(f32x4 x0, f32x4 x1, bf16x8& hi, bf16x8& lo) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        if (AS) {
            u32x4 a = __builtin_bit_cast(u32x4, x0), b = __builtin_bit_cast(u32x4, x1);
            if (RELU) {          // x < 0  <=>  its hi part is negative: sign bit of the word; the whole word becomes +0
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a[c] = (unsigned)max((int)a[c], 0);
                    b[c] = (unsigned)max((int)b[c], 0);
                }
            }
            u32x4 h, l;                       // v_perm_b32: upper / lower halves of two words into one
            h[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302u); h[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302u);
            h[2] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u); h[3] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
            hi = __builtin_bit_cast(bf16x8, h);
            if (PL == 2) {
                l[0] = __builtin_amdgcn_perm(a[1], a[0], 0x05040100u); l[1] = __builtin_amdgcn_perm(a[3], a[2], 0x05040100u);
                l[2] = __builtin_amdgcn_perm(b[1], b[0], 0x05040100u); l[3] = __builtin_amdgcn_perm(b[3], b[2], 0x05040100u);
                lo = __builtin_bit_cast(bf16x8, l);
            }
        } else {
            if (RELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { x0[c] = fmaxf(x0[c], 0.f); x1[c] = fmaxf(x1[c], 0.f); }
            }
            const bf16x4 h0 = __builtin_convertvector(x0, bf16x4), h1 = __builtin_convertvector(x1, bf16x4);
            hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            if (PL == 2) {
                const bf16x4 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f32x4), bf16x4);
                const bf16x4 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f32x4), bf16x4);
                lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    }
    static __device__ __forceinline__ void mma(const char* stage, int wm, int wn, f32x16 (&acc)[TM][TN], int lane, int relu_a = 0) {
        if (relu_a) mma_t<true>(stage, wm, wn, acc, lane);          // wave-uniform: two straight-line bodies, no per-element select
        else mma_t<false>(stage, wm, wn, acc, lane);
    }
    template <bool RELU>
    static __device__ __forceinline__ void mma_t(const char* stage, int wm, int wn, f32x16 (&acc)[TM][TN], int lane) {
        const int li = lane & 31, hi = lane >> 5;
        const float* sA = reinterpret_cast<const float*>(stage) + (wm * TM * 32 + li) * BK;
        const char* sAh = stage + (wm * TM * 32 + li) * BK * 2;
        const char* sW = stage + A_BYTES + (wn * TN * 32 + li) * BK * 2;
        const int swa = (li >> 1) & 7, sww = (li >> 2) & 3;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 a[PL][TM], w[PL][TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                if (AH) {
                    a[0][tm] = frag_half<RELU>(sAh + tm * 32 * BK * 2 + 16 * ((2 * ks + hi) ^ sww));
                } else {
                    const int c0 = (4 * ks + 2 * hi) ^ swa;               // physical chunk of the first four k; the next four sit at c0 ^ 1
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * c0);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(sA + tm * 32 * BK + 4 * (c0 ^ 1));
                    split8<RELU>(x0, x1, a[0][tm], a[PL - 1][tm]);
                }
            }
#pragma unroll
            for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    w[pl][tn] = *reinterpret_cast<const bf16x8*>(sW + pl * W_PLANE + tn * 32 * BK * 2 + 16 * ((2 * ks + hi) ^ sww));
            // term-major order (small terms first): the TM x TN accumulators take turns, so no MFMA waits for the one before it
            if (PL == 2) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = mfma_h<F16>(w[0][tn], a[1][tm], acc[tm][tn]);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = mfma_h<F16>(w[PL - 1][tn], a[0][tm], acc[tm][tn]);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = mfma_h<F16>(w[0][tn], a[0][tm], acc[tm][tn]);
        }
    }
};

template <int BM, int BN, int PREC> struct PipeSel { using type = PipeBF16<BM, BN, PREC>; };
template <int BM, int BN> struct PipeSel<BM, BN, 0> { using type = PipeF32<BM, BN>; };
template <int BM, int BN> struct PipeSel<BM, BN, 4> { using type = PipeF32Dma<BM, BN>; };   // internal: fp32, LDS-direct staging
template <int BM, int BN> struct PipeSel<BM, BN, 5> { using type = PipeSplitDma<BM, BN, 1>; }; // internal: bf16, LDS-direct staging
template <int BM, int BN> struct PipeSel<BM, BN, 7> { using type = PipeSplitDma<BM, BN, 3>; }; // internal: bf16x3, LDS-direct staging
template <int BM, int BN> struct PipeSel<BM, BN, 9> { using type = PipeSplitDma<BM, BN, 1, 1>; };   // ... A in split-pair format
template <int BM, int BN> struct PipeSel<BM, BN, 11> { using type = PipeSplitDma<BM, BN, 3, 1>; };
template <int BM, int BN> struct PipeSel<BM, BN, 13> { using type = PipeSplitDma<BM, BN, 1, 2>; };  // ... A as half rows (bf16)
template <int BM, int BN> struct PipeSel<BM, BN, 15> { using type = PipeSplitDma<BM, BN, 1, 3>; };  // ... A as half rows of fp16, fp16 weight plane

}  // namespace vlsat
