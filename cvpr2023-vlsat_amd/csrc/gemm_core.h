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
#include "common.h"

namespace vlsat {

constexpr int BK = 32;    // k-slice held in LDS per pipeline stage
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
template <int TM, int TN, int PITCH_A = LDT, int PITCH_B = LDT>
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
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][tm][s], b[set][tn][s], acc[tm][tn], 0, 0, 0);
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

}  // namespace vlsat
