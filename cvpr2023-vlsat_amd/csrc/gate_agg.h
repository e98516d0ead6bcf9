// Max aggregation fused into the edge-gate kernels (Aggre_Index with MODEL.GCN_AGGR = max, reference network_util.py:64-73, applied
// to the gated messages of network_MMG.py:104): shared by edge_gate.hip and edge_gate_bf16.hip, whose waves own 32 consecutive
// edges of one head with lane (li = edge row, hi) holding the channels 8 r4 + 4 hi + c of the 32.
#pragma once
#include "common.h"

namespace vlsat {

// max into a float cell from concurrent waves: in the integer order of IEEE bit patterns a non-negative value wins by signed max,
// a negative one by unsigned min (the cell starts at -inf or holds another run's result); exact and order-independent FOR FINITE
// VALUES, where it returns the bits of the separate CSR aggregate kernel.  Corner cases differ and are not promised: a -0.0 / +0.0
// tie resolves to +0.0 here; a NaN from an upstream overflow may win (positive bit pattern), be dropped (negative), or be dropped
// by the per-lane fmaxf pre-reduction, where the aggregate kernel's fmaxf chain drops it always.
__device__ __forceinline__ void atomic_max_f32(float* p, float x) {
    const int b = __float_as_int(x);
    if (b >= 0) atomicMax(reinterpret_cast<int*>(p), b);
    else atomicMin(reinterpret_cast<unsigned*>(p), (unsigned)b);
}

// per wave: a [32 rows][16 channels] fp32 transposition buffer (row pitch 20 floats: the eight rows of a ds_write_b128 lane group
// land on eight different 4-bank groups) and the 32 rows' source nodes
// (each group of eight rows is shifted by 16 more floats, so the two row groups that share an LDS cycle of the column reads below sit
//  on disjoint halves of the 32 banks)
constexpr int AG_PITCH = 20, AG_FLOATS = 32 * AG_PITCH + 3 * 16, AG_WAVE_BYTES = AG_FLOATS * 4 + 32 * 4;
__device__ __forceinline__ int ag_row(int row) { return row * AG_PITCH + (row >> 3) * 16; }

// gated = prob * value for this lane's 16 channels (lg[r] * inv are the probabilities, vrow the value row of the edge's target),
// then the maximum over the rows of each source node WITHOUT storing the gated rows: two passes of 16 channels through the wave's
// LDS buffer `wbuf`; lane (channel lc, row group q) walks rows 8 q .. 8 q + 7 (edge lists are source-major: one or two runs per
// group, but any order is handled) and every finished run goes to agg[src, h*32 + channel] with atomic_max_f32.  sn < 0 marks a
// row past the edge list.
__device__ __forceinline__ void gate_aggregate_max(char* wbuf, const f32x16& lg, float inv, const float* vrow, int sn, int li, int hi,
                                                   int lane, int h, float* agg, int ld_agg) {
    float* tb = reinterpret_cast<float*>(wbuf);
    int* sb = reinterpret_cast<int*>(tb + AG_FLOATS);
    if (hi == 0) sb[li] = sn;
    const int lc = lane & 15, q = lane >> 4;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const i32x4 s03 = *reinterpret_cast<const i32x4*>(sb + 8 * q), s47 = *reinterpret_cast<const i32x4*>(sb + 8 * q + 4);   // sources of this lane's rows
    const int srow[8] = {s03[0], s03[1], s03[2], s03[3], s47[0], s47[1], s47[2], s47[3]};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r4 = 2 * pass + rr;
            const f32x4 v = *reinterpret_cast<const f32x4*>(vrow + 8 * r4);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = lg[r4 * 4 + c] * inv * v[c];
            *reinterpret_cast<f32x4*>(tb + ag_row(li) + 8 * rr + 4 * hi) = o;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (wave-private buffer: program order + this wait)
        // stage 1: the lane's eight rows -> a lead run and a trail run (one and the same when the source never changes: `ls` = -2);
        // runs in between (two or more changes inside eight rows: not with source-major lists) go out at once
        int cur = srow[0], ls = -2;
        float acc = tb[ag_row(8 * q) + lc], la = 0.f;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            const int sj = srow[j];
            const float x = tb[ag_row(8 * q + j) + lc];
            if (sj != cur) {
                if (ls == -2) { ls = cur; la = acc; }
                else if (cur >= 0) atomic_max_f32(agg + (size_t)cur * ld_agg + h * 32 + 16 * pass + lc, acc);
                cur = sj;
                acc = x;
            } else {
                acc = fmaxf(acc, x);
            }
        }
        // stage 2 (round 6): the four row groups of a channel hand their runs to the lane of group 0 through the (now idle) buffer, which
        // merges neighbours of the same source and issues ONE atomic per run of the 32 rows -- one or two per channel and unit on
        // source-major lists instead of five (the atomics leave the XCD: they were 19 % of the bf16 gate's time,
        // profiles/r06_probes/ab_gate_atomics.txt).  max is associative and commutative: the cells end up with the same bits.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (every lane's row reads have returned: the buffer is free)
        *reinterpret_cast<f32x4*>(tb + 4 * lane) = f32x4{__int_as_float(ls), la, __int_as_float(cur), acc};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (q == 0) {
            int rs = -1;
            float ra = 0.f;
            auto take = [&](int sx, float ax) {
                if (sx == rs) ra = fmaxf(ra, ax);
                else {
                    if (rs >= 0) atomic_max_f32(agg + (size_t)rs * ld_agg + h * 32 + 16 * pass + lc, ra);
                    rs = sx;
                    ra = ax;
                }
            };
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 rec = *reinterpret_cast<const f32x4*>(tb + 4 * (lc + 16 * g));
                const int s0 = __float_as_int(rec[0]);
                if (s0 != -2) take(s0, rec[1]);
                take(__float_as_int(rec[2]), rec[3]);
            }
            if (rs >= 0) atomic_max_f32(agg + (size_t)rs * ld_agg + h * 32 + 16 * pass + lc, ra);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the reads are done before the next pass overwrites)
    }
}

}  // namespace vlsat
