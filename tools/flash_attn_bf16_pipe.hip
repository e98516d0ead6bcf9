// (source only, NOT built into the library: round-5 experiment, kept with its measurements -- profiles/r05_probes/flash_pipe.md.
//  To try it again: copy to cvpr2023-vlsat_amd/csrc/, add it to build.py and call launch_flash_attn_bf16_pipe from
//  launch_flash_attn_bf16 for io_split == 2, head dim 64, terms == 1, no split keys.)
// Edge cross-attention core, single-rounding bf16 mode, head dim 64, half-row tensors -- the SOFTWARE-PIPELINED form of
// flash_attn_bf16_kernel<1, true, 2, 3, 64, 2> (flash_attn_bf16.hip; reference transformer/attention.py:60-76 as called from
// network_MMG.py:231).  Same tensors, same LDS images, same lane model (a wave = 32 queries, a lane = one query and one half of
// the keys of a 32-key block, P feeds the second product straight from the score registers, V by ds_read_b64_tr_b16); what
// differs is the ORDER of the work inside a wave.
//
// Why.  Counters of the round-4 kernel at cfg 5 (profiles/r05_probes/flash_pmc_before.txt): per wave and 64-key tile 512 cycles
// of matrix pipe and 672 cycles of VALU (119 instructions: 18 maxima, 32 subtractions, 33 exponentials, 32 additions, 16
// conversions ...), and the time a SIMD spends per wave-tile is their SUM (1168 cycles) -- the matrix pipe is 44 % busy, the VALU
// 57 %, and they are never busy together.  Inside one wave the tile is a dependent chain S = K.Q^T -> softmax -> O += V^T.P, so
// its MFMAs and its VALU cannot overlap; across the four waves of a SIMD they do not overlap either (blocks of equal work run in
// lockstep; start delays and s_setprio changed nothing: profiles/r05_probes/ab_flash_stagger.txt, ab_flash_setprio.txt).
// MFMAs and independent VALU instructions of ONE wave do overlap (MI355X_MICROARCH.md: ~5 single-issue instructions hide under
// each v_mfma_f32_32x32x16_bf16), so the loop is skewed by one tile:
//
//   iteration t:   S(t+1) = K(t+1).Q^T          8 MFMAs   \  issued in between, in four slots of
//                  P(t)   = softmax part of S(t)  VALU     >  [exp of 8 scores per lane | 2 MFMAs of S(t+1) | 2 MFMAs of O += V(t)^T.P(t)]
//                  O     += V(t)^T.P(t)         8 MFMAs   /
//
// K(t+1) and V(t) are live together: K and V tiles sit in two rings of two slots each (tile parity), and iteration t stages
// K(t+2) and V(t+1), each one iteration before it is read.  The loop is unrolled by two, so every slot is a compile-time constant:
// with a run-time slot hipcc cannot tell the LDS-direct writes from the fragment reads and drains vmcnt before the first read of
// an iteration -- the whole memory latency, every tile (that, not occupancy, is what the three- and four-buffer rings of round 4
// lost to).  Plain v_sub / v_add instead of the packed forms (packed fp32
// costs extra beside MFMAs, same guide), v_max3_f32 for the maxima.
// Not built here: split keys (plans of < 512 blocks keep the older kernel), other head dims, split-bf16.
#include <type_traits>

#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

constexpr int FP_KV = 64, FP_D = 64;
constexpr int FP_VSUB = FP_KV * 32 + 128;              // one [64 keys][16 d] sub-tile + half a bank row (as flash_attn_bf16.hip)
constexpr int FP_KBYTES = FP_KV * 128;                 // K image: 64 rows x 128 B, chunks XOR-swizzled with (row >> 1) & 7
constexpr int FP_VPLANE = 4 * FP_VSUB;
constexpr int FP_VBASE = 2 * FP_KBYTES;                // LDS: K slots 0 / 1, then V slots 0 / 1
constexpr int FP_OPITCH = FP_D + 4;
constexpr int FP_SMEM = FP_VBASE + 2 * FP_VPLANE > 4 * 32 * FP_OPITCH * 4 ? FP_VBASE + 2 * FP_VPLANE : 4 * 32 * FP_OPITCH * 4;

typedef short fp_s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N> __device__ __forceinline__ void fp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// max of three without the canonicalising v_max hipcc puts in front of an fmaxf on MFMA results.  Inline asm is invisible to
// the hazard recogniser: every use below reads registers that MFMAs wrote at least half an iteration earlier (or behind fp_settle)
__device__ __forceinline__ float fp_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// MINB: blocks per CU the register budget is set for (3: <= 168 VGPRs, 2: <= 256)
template <int MINB>
__global__ __launch_bounds__(256, MINB) void flash_attn_bf16_pipe_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, float* __restrict__ O,
    int ldq, int ldkv, int ldo, const int4* __restrict__ tiles, int n_tiles) {
    __shared__ __attribute__((aligned(16))) char smem[FP_SMEM];

    const int tile_id = xcd_remap(blockIdx.x, n_tiles);
    const int4 t4 = tiles[tile_id];
    const int row_base = t4.x, n_tok = t4.y, q0 = t4.z, head = t4.w;
    const int n_kv = (n_tok + FP_KV - 1) / FP_KV;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const size_t col0 = (size_t)head * FP_D;
    const bool wave_active = q0 + wave * 32 < n_tok;
    int qrow = q0 + wave * 32 + li;
    if (qrow >= n_tok) qrow = n_tok - 1;               // clamped rows are computed but never stored

    // ---- this lane's query (already multiplied by scale * log2 e by the projection GEMM): d = 16 ks + 8 hi + e ----
    bf16x8 qh[4];
    {
        const char* qp = reinterpret_cast<const char*>(Q + (size_t)(row_base + qrow) * ldq) + (col0 + 8 * hi) * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qh[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
    }

    // ---- LDS-direct staging of a key tile: exactly flash_attn_bf16.hip's (descriptors span this scene: keys past it read as 0) ----
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K + (size_t)row_base * ldkv), 0, (int)(unsigned)((size_t)n_tok * ldkv * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V + (size_t)row_base * ldkv), 0, (int)(unsigned)((size_t)n_tok * ldkv * 4), 0x00020000);
    const unsigned ld4 = (unsigned)ldkv * 4u;
    unsigned vK[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);
        vK[j] = (unsigned)row * ld4 + (unsigned)col0 * 2u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    const unsigned vV0 = (unsigned)(lane >> 1) * ld4 + (unsigned)col0 * 2u + (unsigned)(lane & 1) * 16u;
    auto dma_k = [&](int kv0, char* buf) {
        const unsigned s0 = (unsigned)kv0 * ld4;
#pragma unroll
        for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, buf + (wave * 2 + j) * 1024, 16, vK[j], s0, 0, 0);
    };
    auto dma_v = [&](int kv0, char* buf) {
        const unsigned s0 = (unsigned)kv0 * ld4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = wave * 2 + j, sub = idx >> 1, kh = idx & 1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, buf + sub * FP_VSUB + kh * 1024, 16, vV0, s0 + (unsigned)kh * 32u * ld4 + (unsigned)sub * 32u, 0, 0);
        }
    };

    // ---- fragment addresses (lane constants; the ring slot is added per iteration) ----
    const int kswz = (li >> 1) & 7;
    int offK[4];                                        // K row li (and li + 32 at + 4096), k-step ks
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) offK[ks] = li * 128 + (((hi + 2 * ks) ^ kswz) << 4);
    // V operand of chunk j (keys 16 j + 4 hi + {0..3, 8..11}), d-block db: sub-tile 2 db + ((lane >> 4) & 1)
    // (an LDS byte address: the inline-asm reads below take it as it is)
    const unsigned lds0 = (unsigned)(uintptr_t)((char __attribute__((address_space(3)))*)smem);
    const unsigned offV = lds0 + FP_VBASE + ((lane >> 4) & 1) * FP_VSUB + (4 * hi + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;

    // sa / sb: the score registers of even / odd tiles (the loop is unrolled by two, so "current" and "next" swap by name, not by copy)
    f32x16 o[2], sa[2], sb[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; sa[0][r] = 0.f; sa[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // keys of a partly filled last tile read as zeros: their scores are masked before they enter the maximum
    auto mask_tail = [&](f32x16 (&s)[2], int kv0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kv0 + 32 * kb + crow32(r, hi) >= n_tok) s[kb][r] = -INFINITY;
    };

    // ---- prologue: K(0), V(0), K(1) on their way; S(0) computed without anything to hide behind ----
    dma_k(0, smem);
    dma_v(0, smem + FP_VBASE);
    if (n_kv > 1) { dma_k(FP_KV, smem + FP_KBYTES); fp_wait_vm<2>(); } else fp_wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    if (wave_active) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(smem + offK[ks]);
            const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(smem + offK[ks] + 4096);
            sa[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qh[ks], sa[0], 0, 0, 0);
            sa[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qh[ks], sa[1], 0, 0, 0);
        }
        if (FP_KV > n_tok) mask_tail(sa, 0);
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sa[0]), "+v"(sa[1]));      // MFMA write -> inline-asm VALU read: see fp_max3
    }

    // one key tile; PAR = its parity: V(t) sits in V slot PAR, K(t+1) in K slot PAR ^ 1
    auto tile_step = [&](int t, auto parc, f32x16 (&sc)[2], f32x16 (&sn)[2]) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        const bool has_next = t + 1 < n_kv;
        // K(t+1) and V(t), issued one iteration ago, have landed for everybody, and everybody has left iteration t - 1, which read
        // the slots K(t+2) and V(t+1) now go to
        fp_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 2 < n_kv) dma_k((t + 2) * FP_KV, smem + PAR * FP_KBYTES);
        if (has_next) dma_v((t + 1) * FP_KV, smem + FP_VBASE + (PAR ^ 1) * FP_VPLANE);

        if (wave_active) {
            const char* bK = smem + (PAR ^ 1) * FP_KBYTES;          // K of tile t + 1
            bf16x8 kf[2][2], vf[2];
            auto read_k = [&](int ks, int set) {
                kf[set][0] = *reinterpret_cast<const bf16x8*>(bK + offK[ks]);
                kf[set][1] = *reinterpret_cast<const bf16x8*>(bK + offK[ks] + 4096);
            };
            // V fragments by INLINE-ASM transpose reads.  Through the builtin, hipcc drains vmcnt before the first read of an
            // iteration: it cannot tell an LDS read without alias information from the LDS-direct loads in flight (the tile
            // staged for the NEXT iteration), i.e. the look-ahead would end where the first V fragment is read.  The asm reads are
            // invisible to its wait counting, so the lgkmcnt wait in front of their first use is written out (FP_USE_V: "all but
            // the N newest LDS operations have returned").  Two register sets (A / B) alternate between the 16-key chunks.
            // (macros, not lambdas: hipcc rejects inline-asm operands that name captured variables inside a generic lambda)
            fp_s16x4 vxA0, vxA1, vxA2, vxA3, vxB0, vxB1, vxB2, vxB3;
            const unsigned vaddr = offV + (unsigned)(PAR * FP_VPLANE);
#define FP_READ_V(J, S)                                                                                                       \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"                              \
                 "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"                                   \
                 : "=&v"(vx##S##0), "=&v"(vx##S##1), "=&v"(vx##S##2), "=&v"(vx##S##3)                                         \
                 : "v"(vaddr), "n"((J) * 512), "n"((J) * 512 + 256), "n"((J) * 512 + 2 * FP_VSUB), "n"((J) * 512 + 2 * FP_VSUB + 256))
#define FP_USE_V(S, N)                                                                                              \
    do {                                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(vx##S##0), "+v"(vx##S##1), "+v"(vx##S##2), "+v"(vx##S##3));   \
        vf[0] = __builtin_shufflevector(__builtin_bit_cast(bf16x4, vx##S##0), __builtin_bit_cast(bf16x4, vx##S##1), 0, 1, 2, 3, 4, 5, 6, 7); \
        vf[1] = __builtin_shufflevector(__builtin_bit_cast(bf16x4, vx##S##2), __builtin_bit_cast(bf16x4, vx##S##3), 0, 1, 2, 3, 4, 5, 6, 7); \
    } while (0)
#define FP_SB() __builtin_amdgcn_sched_barrier(0)
            // sched_barrier orders the machine scheduler only; the passes before it still sink a pure MFMA or conversion towards its
            // use (they all ended up behind the last exponential).  An empty volatile asm that names the result pins the producer
            // in front of it, and volatile asms keep their program order.
#define FP_PIN(x) asm volatile("" : "+v"(x))
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // ONE MFMA at a time, each between two groups of ~8 VALU instructions (a wave issues in order: MFMAs placed back to
            // back wait for the pipe -- 8 issue slots each -- while the VALU work that could have filled them queues up behind).
            // (S(t+1) is computed unconditionally: in the last iteration the slot holds an older tile and the result is dropped --
            //  a branch around the MFMAs made hipcc copy the 32 score registers at each of its join points)
            auto qk = [&](int ks, int set, int half) {     // S(t+1)[32 half ..] += K(t+1)[32 half .., 16 ks ..] . Q^T  (k-step 0 starts from the constant 0)
                sn[half] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[set][half], qh[ks], ks == 0 ? zero : sn[half], 0, 0, 0);
                FP_PIN(sn[half]);
            };
            bf16x8 ph[4];
            f32x4 p0, p1;
            float m_use = 0.f;
            f32x2 rsv0 = {0.f, 0.f}, rsv1 = {0.f, 0.f}, mm = {0.f, 0.f};
            // score - m and the row sums as fp32 PAIRS (v_pk_add_f32: one issue slot for two values; the counters of the first
            // version of this kernel, which used scalar adds, showed +26 % VALU time for the same work)
            auto e0 = [&](int j) {                         // exponentials of scores 8 (j & 1) .. + 3 of key block j >> 1
                const int kb = j >> 1, i = 8 * (j & 1);
                const f32x2 a = f32x2{sc[kb][i], sc[kb][i + 1]} - mm, b = f32x2{sc[kb][i + 2], sc[kb][i + 3]} - mm;
                p0 = f32x4{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1]), __builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
                FP_PIN(p0);
            };
            auto e1 = [&](int j) {
                const int kb = j >> 1, i = 8 * (j & 1) + 4;
                const f32x2 a = f32x2{sc[kb][i], sc[kb][i + 1]} - mm, b = f32x2{sc[kb][i + 2], sc[kb][i + 3]} - mm;
                p1 = f32x4{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1]), __builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
                FP_PIN(p1);
            };
            auto cv = [&](int j) {                         // row-sum contribution and the bf16 operand of chunk j
                rsv0 += f32x2{p0[0], p0[1]} + f32x2{p0[2], p0[3]};
                rsv1 += f32x2{p1[0], p1[1]} + f32x2{p1[2], p1[3]};
                ph[j] = __builtin_shufflevector(__builtin_convertvector(p0, bf16x4), __builtin_convertvector(p1, bf16x4), 0, 1, 2, 3, 4, 5, 6, 7);
                FP_PIN(ph[j]);
                FP_PIN(rsv0);
                FP_PIN(rsv1);
            };
            auto pv = [&](int j, int db) {
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db], ph[j], o[db], 0, 0, 0);
                FP_PIN(o[db]);
            };

            read_k(0, 0);
            read_k(1, 1);
            // ---- 16 VALU groups of 8-10 issue slots, one MFMA behind each ----
            // groups 1-4: maximum of the tile's 32 scores of this lane, running maximum, rescale; MFMAs 1-4 = k-steps 0, 1 of S(t+1)
            float ma = fp_max3(sc[0][0], sc[0][1], sc[0][2]), mb = fp_max3(sc[1][0], sc[1][1], sc[1][2]);
#pragma unroll
            for (int r = 3; r < 9; r += 2) { ma = fp_max3(ma, sc[0][r], sc[0][r + 1]); mb = fp_max3(mb, sc[1][r], sc[1][r + 1]); }
            FP_SB(); qk(0, 0, 0); FP_SB();
#pragma unroll
            for (int r = 9; r < 15; r += 2) { ma = fp_max3(ma, sc[0][r], sc[0][r + 1]); mb = fp_max3(mb, sc[1][r], sc[1][r + 1]); }
            float mx = fp_max3(ma, mb, sc[0][15]);
            mx = fp_max3(mx, sc[1][15], sc[1][15]);
            FP_SB(); qk(0, 0, 1); read_k(2, 0); FP_SB();
            mx = fp_max3(mx, __shfl_xor(mx, 32), mx);
            const float m_new = fmaxf(m_run, mx);
            m_use = m_new == -INFINITY ? 0.f : m_new;
            mm = f32x2{m_use, m_use};
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            m_run = m_new;
            FP_SB(); qk(1, 1, 0); FP_SB();
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {           // (behind the previous tile's last O MFMA, before this tile's first)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            }
            FP_SB(); qk(1, 1, 1); read_k(3, 1); FP_READ_V(0, A); FP_SB();
            // groups 5-16: the four 16-key chunks; MFMAs 5-8 = k-steps 2, 3 of S(t+1), 9-16 = O += V(t)^T . P(t) one chunk behind
            e0(0); FP_SB(); qk(2, 0, 0); FP_SB();
            e1(0); FP_SB(); qk(2, 0, 1); FP_SB();
            cv(0); FP_SB(); qk(3, 1, 0); FP_SB();
            e0(1); FP_SB(); qk(3, 1, 1); FP_READ_V(1, B); FP_SB();
            e1(1); FP_SB(); FP_USE_V(A, 4); pv(0, 0); FP_SB();
            cv(1); FP_SB(); pv(0, 1); FP_READ_V(2, A); FP_SB();
            e0(2); FP_SB(); FP_USE_V(B, 4); pv(1, 0); FP_SB();
            e1(2); FP_SB(); pv(1, 1); FP_READ_V(3, B); FP_SB();
            cv(2); FP_SB(); FP_USE_V(A, 4); pv(2, 0); FP_SB();
            e0(3); FP_SB(); pv(2, 1); FP_SB();
            e1(3); FP_SB();
            cv(3); FP_SB(); FP_USE_V(B, 0); pv(3, 0); FP_SB();
            float rs = (rsv0[0] + rsv0[1]) + (rsv1[0] + rsv1[1]);
            rs += __shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            FP_SB(); pv(3, 1); FP_SB();
            if (has_next && (t + 2) * FP_KV > n_tok) mask_tail(sn, (t + 1) * FP_KV);
        }
    };
#undef FP_READ_V
#undef FP_USE_V
#undef FP_SB
#undef FP_PIN
    for (int t = 0; t < n_kv; t += 2) {
        tile_step(t, std::integral_constant<int, 0>{}, sa, sb);
        if (t + 1 < n_kv) tile_step(t + 1, std::integral_constant<int, 1>{}, sb, sa);
    }

    // ---- normalise, transpose through LDS (wave-private [32 q][68]), coalesced half-row store ----
    __syncthreads();                                    // (the tile buffers are dead: every wave has left the loop)
    const float inv_l = 1.f / l_run;
    float* so = reinterpret_cast<float*>(smem) + wave * (32 * FP_OPITCH);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) so[li * FP_OPITCH + 32 * b + crow32(r, hi)] = o[b][r] * inv_l;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FP_D / 8; ++i) {
        const int idx = lane + 64 * i;
        const int r = idx / (FP_D / 4), c4 = (idx % (FP_D / 4)) * 4;
        const int qr = q0 + wave * 32 + r;
        if (qr < n_tok) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(so + r * FP_OPITCH + c4);
            float* orow = O + (size_t)(row_base + qr) * ldo;
            *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(orow) + (col0 + c4) * 2) = __builtin_convertvector(v, bf16x4);
        }
    }
}

}  // namespace

// half rows, single rounding, head dim 64, no split keys; the caller has checked the 32-bit range of a scene's rows
int launch_flash_attn_bf16_pipe(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo,
                                const int4* tiles, int n_tiles, hipStream_t s, int variant) {
    if (n_tiles <= 0) return 0;
    if (variant == 2) hipLaunchKernelGGL(flash_attn_bf16_pipe_kernel<2>, dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles);
    else hipLaunchKernelGGL(flash_attn_bf16_pipe_kernel<3>, dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles);
    VLSAT_LAUNCH_CHECK("flash_attn_bf16_pipe");
    return 0;
}

}  // namespace vlsat
