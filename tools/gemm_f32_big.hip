// Large-M fp32 MFMA GEMM, "one wave per SIMD" variant -- EXPERIMENTAL, opt-in (vlsat_debug_gemm_variant).
// Operand layouts, the persistent XCD-aware tile walk and the epilogue semantics are those of gemm_f32.hip.
//
//   block  = 256 threads = 4 waves (2 x 2), ONE block per CU -> each wave owns a SIMD and its matrix pipe
//   tile   = 256 x 128, wave tile 128 x 64 = 4 x 2 MFMA tiles (128 accumulator registers; the whole 512-entry
//            register file of the SIMD belongs to this wave)
//   k-step = 16 (two k-groups of 8): 64 MFMAs = 4096 pipe cycles per wave per step
//   LDS    = ring of FOUR stages of [256+128][16] fp32 (row pitch 20 floats, conflict-free ds_read_b128);
//            global loads move 32-k slabs (full 128-byte lines) = two stages at a time
//
// Idea: with two 4-wave blocks per CU (gemm_f32.hip) the matrix pipe measured 85-88 % busy (in-kernel
// timers): after every barrier a wave waits for LDS data that was written just before that barrier.  Here
// slab T+1 is written during the first step of slab T, so it is visible one whole step before it is read; the
// fragment reads of step t+1 are issued BEFORE the barrier that ends step t, the barrier is a raw s_barrier
// behind a counted lgkmcnt wait (no vmcnt drain: global loads stay in flight across it), and because a lone
// wave has nobody to hide its non-MFMA instructions behind, every step is 16 fenced chunks of 4 MFMAs that
// each carry one LDS store + one global load.
//
// Measured (tools/gemm_step_probe.py, tools/gemm_bench.py): lost cycles = ~10 k per tile +
// ~460 per step, i.e. 84-87 % pipe busy; 106.7 / 112.8 / 115.0 TFLOP/s on E x512x512 / E x1024x512 /
// E x512x1024 against 99 / 106 / 110 for the default kernel and 103.5 / 125.6 / 127.2 for the vendor BLAS
// (tools/blas_reference.py).  Ablations: the global loads cost ~270 of the 460 cycles per step (VMEM issue
// from the only wave of the SIMD), LDS stores ~65, the barrier ~60.  Launches with additive operands are
// slower than the default (their loads are exposed at the tile start), and in the full forward the variant
// is a wash (1961 vs 1955 scenes/s), so it stays off by default.  Next step if revisited: LDS-direct global
// loads (no VGPR round trip, no ds_write) with an XOR-swizzled unpadded layout.
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {
constexpr int GB_BM = 256, GB_BN = 128, GB_TM = 4, GB_TN = 2;
constexpr int GB_KS = 16;                       // k per step
constexpr int GB_PITCH = 20;                    // floats per LDS row (16 + 4 pad)
constexpr int GB_STAGE = (GB_BM + GB_BN) * GB_PITCH + 16;  // floats per stage; +16 so that the two stages a
                                                           // 128-byte row is split over start 16 banks apart
constexpr int GB_NS = 4;

struct GbRegs { f32x4 r[12]; };                 // one 32-k slab: 384 rows x 8 float4 / 256 threads
struct GbFrag { f32x4 a[GB_TM], b[GB_TN]; };

}  // namespace

template <int ADD, bool RELU_A>
__global__ __launch_bounds__(256, 1) void gemm_f32_big_kernel(GemmArgs p, int n_tiles, int nbn) {
    constexpr int TM = GB_TM, TN = GB_TN, BM = GB_BM, BN = GB_BN;
    __shared__ __attribute__((aligned(16))) float smem[GB_NS * GB_STAGE];      // 123 136 B

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hi = lane >> 5;
    const int g8 = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int KT = p.K / GB_KS;                                                 // steps per tile (even, >= 4)
    auto tile_of = [&](int round) { return (round * 8 + xcd) * g8 + slot; };

    const int v0 = tile_of(0);
    if (v0 >= n_tiles) return;
    const long long t_start = p.clock_probe ? clock64() : 0, w_start = p.clock_probe ? wall_clock64() : 0;

    // ---- load cursor: runs two slabs (four steps) ahead of the compute cursor; past the last tile it stays on the
    //      last tile (the extra loads are never consumed), so the step body has no branches ----
    int lm0 = (v0 / nbn) * BM, ln0 = (v0 % nbn) * BN, lk = 0, lround = 0;
    // staging map of one 32-k slab (= two ring stages): thread -> (row, 16-byte column); 8 consecutive lanes
    // fetch one full 128-byte line of a row, the first 64 bytes go to the even stage, the rest to the odd one.
    // Piece j covers rows 32 j + tid/8 (j < 8: A rows, else W rows).  Row offsets (in floats, clamped to the
    // matrix) live in VGPRs and are recomputed when the cursor moves to another tile.
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    unsigned off[12];
    auto set_offsets = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int row = lm0 + srow + 32 * j;
            row = row < p.M - 1 ? row : p.M - 1;
            off[j] = (unsigned)row * (unsigned)p.lda + scol;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = ln0 + srow + 32 * j;
            row = row < p.N - 1 ? row : p.N - 1;
            off[8 + j] = (unsigned)row * (unsigned)p.ldw + scol;
        }
    };
    set_offsets();
    auto advance = [&]() {                              // one slab (two steps) ahead
        lk += 2 * GB_KS;
        const bool wrap = lk == p.K;
        lk = wrap ? 0 : lk;
        lround += wrap ? 1 : 0;
        const int nv = tile_of(lround);
        const bool mv = wrap && nv < n_tiles;
        if (mv) {
            lm0 = (nv / nbn) * BM;
            ln0 = (nv % nbn) * BN;
            set_offsets();
        }
    };
    auto load_piece = [&](GbRegs& r, int j) {
        r.r[j] = *reinterpret_cast<const f32x4*>((j < 8 ? p.A : p.W) + lk + off[j]);
    };
    // `stage` = the even stage of the slab's pair
    auto store_piece = [&](float* stage, const GbRegs& r, int j) {
        float* d = stage + (scol >> 4) * GB_STAGE + (srow + 32 * j) * GB_PITCH + (scol & 15);
        f32x4 x = r.r[j];
        if (RELU_A && j < 8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = fmaxf(x[c], 0.f);
        }
        *reinterpret_cast<f32x4*>(d) = x;
    };
    auto load = [&](GbRegs& r) {
#pragma unroll
        for (int j = 0; j < 12; ++j) load_piece(r, j);
    };
    auto store = [&](float* stage, const GbRegs& r) {
#pragma unroll
        for (int j = 0; j < 12; ++j) store_piece(stage, r, j);
    };
    const int fa_off = (wm * TM * 32 + li) * GB_PITCH + hi * 4;
    const int fb_off = (BM + wn * TN * 32 + li) * GB_PITCH + hi * 4;
    auto frag = [&](const float* stage, int kg, GbFrag& f) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) f.a[tm] = *reinterpret_cast<const f32x4*>(stage + fa_off + tm * 32 * GB_PITCH + kg * 8);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) f.b[tn] = *reinterpret_cast<const f32x4*>(stage + fb_off + tn * 32 * GB_PITCH + kg * 8);
    };

    GbRegs regs;
    // prologue: slab 0 (steps 0, 1) into stages 0, 1; slab 1 into registers (K >= 64: same tile)
    load(regs); advance();
    store(smem, regs);
    __syncthreads();
    load(regs); advance();

    int round = 0;
    int m0 = lm0, n0 = ln0;                            // K >= 64: the cursor has not left the first tile...
    m0 = (v0 / nbn) * BM; n0 = (v0 % nbn) * BN;        // ...but K == 64 wraps exactly here: recompute
    f32x16 acc[TM][TN];

    // accumulator init of tile (tm0, tn0): zero or the additive operands (residual / gathered rows)
    auto init_acc = [&](int tm0, int tn0) {
        if (ADD == 0) { zero_acc<TM, TN>(acc); return; }
        int ldr = p.ldr, ldg0 = p.ldg0, ldg1 = p.ldg1, lv = lane;
        asm volatile("" : "+s"(ldr), "+s"(ldg0), "+s"(ldg1), "+v"(lv));   // no LICM of per-lane offsets
        const int eli = lv & 31, ehi = lv >> 5;
        const float* rbase = (ADD & 1) ? p.resid + (size_t)tm0 * ldr + tn0 : nullptr;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            int nl = (wn * TN + tn) * 32 + eli;
            if (tn0 + nl >= p.N) nl = p.N - 1 - tn0;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int ml = (wm * TM + tm) * 32 + crow32(r, ehi);
                    if (tm0 + ml >= p.M) ml = p.M - 1 - tm0;
                    float x = 0.f;
                    if (ADD & 1) x = p.resid_scale * rbase[(unsigned)(ml * ldr + nl)];
                    if (ADD & 2) x += p.g0[(unsigned)(p.gi0[tm0 + ml] * ldg0 + tn0 + nl)];
                    if (ADD & 4) x += p.g1[(unsigned)(p.gi1[tm0 + ml] * ldg1 + tn0 + nl)];
                    acc[tm][tn][r] = x;
                }
        }
    };
    init_acc(m0, n0);

    GbFrag fa, fb;
    frag(smem, 0, fa);                                 // k-group 0 of step 0

    int gstep = 0;                                      // global step counter of this block (ring index)
    while (true) {
        const int nv = tile_of(round + 1);
        const bool next_tile = nv < n_tiles;
        const int nm0 = (nv / nbn) * BM, nn0 = (nv % nbn) * BN;
        // One k-step, branch-free so that it is ONE scheduling region.  `regs` holds the data of step t+2 on
        // entry and the loads of step t+4 on exit.  A single wave per SIMD has nobody to hide its non-MFMA
        // instructions behind, so they are interleaved between the 64 MFMAs explicitly (each MFMA keeps
        // the pipe busy for 64 cycles and needs 4-8 to issue).
        auto step = [&](const bool even) {
            float* cur = smem + (gstep & 3) * GB_STAGE;
            float* nx1 = smem + ((gstep + 1) & 3) * GB_STAGE;
            float* wr = smem + ((gstep + 2) & 3) * GB_STAGE;      // even step: first stage of the slab after next
            __builtin_amdgcn_sched_barrier(0);
            frag(cur, 1, fb);                           // second k-group of this step
            __builtin_amdgcn_sched_barrier(0);
            // 16 chunks of 4 MFMAs, fenced, each carrying a small piece of the other work:
            //   even step, chunks 0..11: LDS store of piece c of the NEXT slab (loaded two steps ago), then the
            //                            global load of piece c of the slab after it into the same register
            //   after chunk 7          : fragment reads of step t+1's first k-group (visible since the last barrier)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (even && c < 12) {
                    store_piece(wr, regs, c);
                    load_piece(regs, c);
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int q = (c & 7) * 4 + q4, ks = q >> 3, tm = (q >> 1) & 3, tn = q & 1;
                    if (c < 8) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.a[tm][ks], fa.b[tn][ks], acc[tm][tn], 0, 0, 0);
                    else acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.a[tm][ks], fb.b[tn][ks], acc[tm][tn], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (c == 7) {
                    frag(nx1, 0, fa);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // all LDS writes of this step are done (fragment reads issued after them may still be in flight:
            // 6 in an odd step, none younger than the last 4 stores in an even one); the global loads stay in
            // flight across the barrier
            if (even) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            ++gstep;
            if (even) advance();
        };
        for (int kt = 0; kt < KT; kt += 2) {            // K % 32 == 0: an even number of steps
            step(true);
            step(false);
        }
        // ---- epilogue of tile (m0, n0) ----
        {
            int ldc = p.ldc, lv = lane;
            asm volatile("" : "+s"(ldc), "+v"(lv));
            const int eli = lv & 31, ehi = lv >> 5;
            float* cbase = p.C + (size_t)m0 * ldc + n0;
            if (p.rowscale) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int m = m0 + (wm * TM + tm) * 32 + crow32(r, ehi);
                        m = m < p.M ? m : p.M - 1;
                        const float rs = p.rowscale[m];
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) acc[tm][tn][r] *= rs;
                    }
            }
            if (p.bias) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    int n = n0 + (wn * TN + tn) * 32 + eli;
                    n = n < p.N ? n : p.N - 1;
                    const float bn = p.bias[n];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[tm][tn][r] += bn;
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
            if (m0 + BM <= p.M && n0 + BN <= p.N) {          // interior tile: branch-free stores
                float* c0 = cbase + (wm * TM * 32 + 4 * ehi) * ldc + wn * TN * 32 + eli;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        float* crow = c0 + (unsigned)((tm * 32 + 8 * r4) * ldc);
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) crow[(unsigned)(rr * ldc) + tn * 32] = acc[tm][tn][r4 * 4 + rr];
                    }
            } else {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ml = (wm * TM + tm) * 32 + crow32(r, ehi), nl = (wn * TN + tn) * 32 + eli;
                            if (m0 + ml < p.M && n0 + nl < p.N) cbase[(unsigned)(ml * ldc + nl)] = acc[tm][tn][r];
                        }
            }
        }
        if (!next_tile) {
            if (p.clock_probe && tid == 0) {
                long long* d = p.clock_probe + (size_t)blockIdx.x * 4;
                d[0] = clock64() - t_start; d[1] = wall_clock64() - w_start; d[2] = round + 1; d[3] = 1;
            }
            break;
        }
        ++round;
        m0 = nm0;
        n0 = nn0;
        init_acc(m0, n0);
    }
}

// tiles [0, n_tiles) of the 256 x 128 tiling of `a` (N fastest) on a persistent grid of `grid` blocks (1 per CU)
int launch_gemm_big(const GemmArgs& a, int n_tiles, int grid, hipStream_t s) {
    if (a.K < 64 || a.K % (2 * GB_KS)) return 1;
    const int nbn = (a.N + GB_BN - 1) / GB_BN;
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
    if ((size_t)a.M * a.lda >= (1ull << 31) || (size_t)a.N * a.ldw >= (1ull << 31)) return 1;   // 32-bit row offsets
    const int key = add * 2 + (a.relu_a ? 1 : 0);
    switch (key) {
        case 0: hipLaunchKernelGGL((gemm_f32_big_kernel<0, false>), dim3(grid), dim3(256), 0, s, a, n_tiles, nbn); break;
        case 1: hipLaunchKernelGGL((gemm_f32_big_kernel<0, true>), dim3(grid), dim3(256), 0, s, a, n_tiles, nbn); break;
        case 2: hipLaunchKernelGGL((gemm_f32_big_kernel<1, false>), dim3(grid), dim3(256), 0, s, a, n_tiles, nbn); break;
        case 12: hipLaunchKernelGGL((gemm_f32_big_kernel<6, false>), dim3(grid), dim3(256), 0, s, a, n_tiles, nbn); break;
        default: return 1;     // combination not built: the caller uses the 4-wave kernel
    }
    VLSAT_LAUNCH_CHECK("gemm_f32_big");
    return 0;
}

}  // namespace vlsat
