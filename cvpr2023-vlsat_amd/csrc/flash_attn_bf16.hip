// Edge cross-attention core on the bf16 matrix cores (BASELINE configs[2]): the same flash-style kernel as
// flash_attn_f32.hip -- reference transformer/attention.py:60-76 as called from network_MMG.py:231, scores never
// leave registers -- with QK^T and PV on v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate) and fp32 softmax.
//
// TERMS = 3 (split-bf16, ~1e-5): every operand x is carried as bf16 hi + bf16 lo (x ~= hi + lo to 2^-17) and each
// product is three MFMAs  a_lo.b_hi + a_hi.b_lo + a_hi.b_hi  (small terms first), fp32 accumulate.
// TERMS = 1 (single rounding): hi parts only.
//
// Q, K, V arrive as fp32 (the projection GEMMs' outputs) and are split while staging: Q once per block into registers
// (pre-multiplied by scale*log2 e), K/V per key tile into LDS planes.
//   block   = 128 queries (4 waves x 32), key tiles of 64, double-buffered LDS, one barrier per tile;
//   S^T     = K_tile . Q^T  : A = K rows from LDS (ds_read_b128: 8 consecutive d, row pitch 144 B = conflict-free),
//             B = Q from registers; "swapped" product so a lane owns ONE query column: the softmax row reductions
//             are in-lane plus one lane^32 shuffle;
//   O^T    += V^T . P^T     : B = P straight from the S registers -- the k-slot (half hi, element e) of step j is key
//             32(j>>1) + 16(j&1) + 8(e>>2) + 4 hi + (e&3), i.e. exactly the keys registers 8(j&1)+e already hold, so P
//             needs no cross-lane movement; A = V^T[d][those keys] read from the ROW-major V tile with the gfx950
//             LDS transpose read ds_read_b64_tr_b16 (two reads of 4 keys x 16 d per operand).  The V image is
//             4 sub-tiles [64 keys][16 d] (32-byte rows) 2176 B apart, so the two 16-lane groups of a half-wave land
//             on disjoint bank halves (semantics and bank behaviour: tools/tr_read_probe.hip).
//   TR = false keeps a gather fallback for the V operand (eight ds_read_u16 per operand) -- the reference
//   implementation the transpose-read path is tested against.
// Split keys (plans that cannot fill the chip) and the output transpose through LDS are as in flash_attn_f32.hip.
// Roofline: bf16 MFMA, 2.5 PF / TERMS; algorithmic work 4*T^2*64 flop per (scene, head).
#include <type_traits>

#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

constexpr int FB_KV = 64;              // keys per tile
constexpr int FB_VSUB = FB_KV * 32 + 128;                     // 2176: one [64 keys][16 d] sub-tile + half a bank row
// FB_D (template): head dim = 512 / MODEL.NUM_HEADS: 64 as shipped, 32 or 128 for 16 / 4 heads (round 3).  It sets the number
// of 16-wide k-steps of QK^T (FB_D / 16), of 32-wide output blocks (FB_D / 32) and of V sub-tiles (FB_D / 16); K rows are
// 2 FB_D + 16 bytes apart (80 / 144 / 272: every 16-lane group of a ds_read_b128 lands on 16 different 4-bank groups).

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void fb_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// max of three without the canonicalising v_max hipcc puts in front of an fmaxf on MFMA results
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// packed fp32 pairs: one full-rate VALU instruction for two values.  Inline asm is invisible to hipcc's hazard recogniser: on
// gfx950 a VALU instruction that reads the result of a transcendental one (v_exp_f32) needs a wait state in between, which the
// compiler inserts only before instructions it knows to be VALU -- hence the s_nop in front of the add that takes the exponentials
// (found by the kernel tests: without it the row sums used stale registers).
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, float m) {      // {a.x - m, a.y - m}
    f32x2 d;
    const f32x2 mm = {m, m};
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(mm));
    return d;
}

__device__ __forceinline__ void split4(const f32x4& x, bf16x4& hi, bf16x4& lo) {
    hi = __builtin_convertvector(x, bf16x4);
    lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), bf16x4);
}
// four elements of an operand tensor -> bf16 hi / lo.  IO_S: the tensor is in the split-pair format (common.h
// pack_split; written by the projection GEMMs' epilogues in the bf16 modes), so this is two v_perm_b32 per plane
template <int IO>
__device__ __forceinline__ void planes4(const f32x4& x, bf16x4& hi, bf16x4& lo) {
    if (IO == 1) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x4 w = __builtin_bit_cast(u32x4, x);
        u32x2 h, l;
        h[0] = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u); h[1] = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
        l[0] = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u); l[1] = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u);
        hi = __builtin_bit_cast(bf16x4, h);
        lo = __builtin_bit_cast(bf16x4, l);
    } else {
        split4(x, hi, lo);
    }
}

// four consecutive elements of a row, whatever the tensor format: IO = 0 fp32, 1 split-pair words (16 bytes each way),
// 2 half rows (8 bytes: four bf16, returned in the low half of the f32x4)
template <int IO>
__device__ __forceinline__ f32x4 load4(const float* row, size_t col) {
    if (IO >= 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 v = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(row) + col * 2);
        return f32x4{v[0], v[1], 0.f, 0.f};
    }
    return *reinterpret_cast<const f32x4*>(row + col);
}
template <int IO>
__device__ __forceinline__ void planes_of(const f32x4& x, bf16x4& hi, bf16x4& lo) {
    if (IO >= 2) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        hi = __builtin_bit_cast(bf16x4, f32x2{x[0], x[1]});
        lo = hi;                                   // (never read: TERMS = 1)
    } else {
        planes4<IO>(x, hi, lo);
    }
}

// IO = 1 / 2: Q, K, V arrive and O leaves in the split-pair / half-row format; Q is then already multiplied by
// scale * log2 e (the projection GEMM's epilogue did it)
// PVT (split-bf16 mode only): MFMAs per P.V product.  3 = V_hi.P_hi + V_lo.P_hi + V_hi.P_lo; 2 drops the last term, i.e. the
// probabilities enter the second product single-rounded (V stays exact) -- an experiment switch (flash_pv_terms).
// DMA (round 4; half rows, single rounding, head dim 64): K / V tiles arrive by LDS-direct loads (buffer_load ... lds, no VGPR
// round trip, no ds_write) into a ring of RING tile buffers (2 as shipped: one tile ahead, four blocks per CU), RING - 1 tiles
// ahead of the one being computed, with counted s_waitcnt vmcnt and a raw s_barrier per tile.  Measured before: with the staging loads switched off the kernel ran 28 %
// faster, with the LDS stores off as well 43 % (profiles/r04_probes/flash_bf16_staging_ablation.txt) -- one tile of look-ahead
// through registers does not cover an L2 / HBM round trip inside a 0.45 us iteration.  The K image is then 64 rows x 128 B with
// the 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the global side and on the fragment reads (an LDS-direct load writes
// lane-linear, so the 144-byte row pitch of the register-staged image cannot be produced); the V image is unchanged (four
// [64 keys][16 d] sub-tiles: a wave instruction fills 32 keys x 32 B of one of them).
// BQW: waves = 32-query groups per block: 4 (128 queries, the default) or -- LDS-direct staging only -- 8 (256 queries sharing every
// K / V tile: half the L2 -> LDS bytes per query; for scenes of thousands of tokens, whose K / V no longer sit in one XCD's L2 and
// whose last, partly filled query tile is a negligible share: round 5, +4 % at cfg 5)
// ABL (experiments build only, compile-time so that the shipped code generation is what gets timed; results are GARBAGE): 4 no exponentials,
// 8 no maximum, 16 no cross-half exchanges, 32 no P.V MFMAs, 64 no Q.K MFMAs, 256 no barrier per tile
// (round 6, measured and dropped: the V fragments by inline-asm ds_read_b64_tr_b16 with hand-counted lgkmcnt waits, two register sets a
//  16-key chunk apart -- through the builtin, which carries no alias information, hipcc drains vmcnt in front of the first transpose read
//  of an iteration, i.e. a wave waits for the NEXT tile's LDS-direct loads where its P.V product starts.  Bit-identical, 120 VGPRs, still
//  four blocks per CU, no vmcnt wait left inside the iteration: 9844-9894 vs 9822-9870 scenes/s in the cfg 3 step (+0.2 %), 80.4-81.0 vs
//  80.7-81.7 at cfg 5 (-0.7...-1.2 %): with K / V resident in L2 the loads have landed by then, and the fixed read order costs more than
//  the compiler's.  profiles/r06_probes/krot_asmv_step_ab.txt)
// QG (round 6): 32-query groups per WAVE.  2 = a wave owns 64 queries: every K and V fragment it reads from LDS feeds
// two MFMAs (half the fragment reads, address arithmetic, waits and loop control per product), Q and O of both groups stay in
// registers (~200 VGPRs: two waves per SIMD where QG = 1 has four of 120).  A query's arithmetic is the same instruction sequence on
// the same keys in the same order, so the outputs equal QG = 1 BIT FOR BIT (tests/test_hip_round6.py).  Block = BQW waves x 32 QG
// queries: (BQW 2, QG 2) serves the 128-query tile table, (4, 2) the 256-query one.  ORD 1: the P.V product of group 0 is issued
// in front of group 1's softmax (its V fragments are then read twice).
template <int TERMS, bool TR, int IO, int PVT = 3, int FB_D = 64, int RING = 0, int BQW = 4, int ABL = 0, int QG = 1, int ORD = 0>
__global__ __launch_bounds__(64 * BQW, 2) void flash_attn_bf16_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, int ldq, int ldkv, int ldo, const int4* __restrict__ tiles, int n_tiles,
    float scale_log2e, FlashSplit sp) {
    constexpr int PL = TERMS == 1 ? 1 : 2;
    constexpr int NKS = FB_D / 16, NO = FB_D / 32, NSUB = FB_D / 16;
    constexpr int FB_KPITCH = 2 * FB_D + 16;       // bytes per key row of a K plane (conflict-free ds_read_b128)
    constexpr int FB_KPLANE = FB_KV * FB_KPITCH;
    constexpr int FB_VPLANE = NSUB * FB_VSUB;
    constexpr int FB_OPITCH = FB_D + 4;            // floats per query row of the output transpose
    constexpr bool DMA = RING != 0;           // RING: tile buffers of the LDS-direct K/V ring (0: register-staged double buffer)
    static_assert(!DMA || (TERMS == 1 && IO >= 2 && TR), "LDS-direct K/V staging: half rows, single rounding");
    constexpr bool F16 = IO == 3;                  // IO 3: the half rows hold fp16, QK^T and PV run on the f16 MFMA (precision mode fp16_mixed)
    static_assert(!F16 || (TERMS == 1 && DMA), "fp16 half rows: the LDS-direct single-rounding kernel");
    // (DMA) K image: 64 rows of ROWB = 2 FB_D bytes, unpadded (an LDS-direct load writes lane-linear), 16-byte chunks XOR-swizzled with
    // KSWZ(row) on the global side and on the fragment reads: 64-byte rows (row >> 2) & 3, 128-byte rows (row >> 1) & 7, 256-byte rows row & 15
    constexpr int ROWB = 2 * FB_D, CPR = ROWB / 16, RPI = 1024 / ROWB;          // bytes per K row, chunks per row, rows per load instruction
    static_assert(QG == 1 || (QG == 2 && RING != 0 && FB_D == 64 && (BQW == 2 || BQW == 4)), "two query groups per wave: the LDS-direct variant at head dim 64");
    static_assert(BQW == 4 || (BQW == 8 && RING != 0 && FB_D == 64) || (BQW == 2 && QG == 2), "eight waves: the LDS-direct variant at head dim 64");
    constexpr int KI = FB_KV / RPI / BQW, VI = 2 * NSUB / BQW, LPT = KI + VI;      // load instructions per wave and tile: K, V, both
    constexpr int KBYTES = FB_KV * ROWB;
    constexpr int BUF = DMA ? KBYTES + FB_VPLANE : PL * (FB_KPLANE + FB_VPLANE);
    constexpr int NBUF = DMA ? RING : 2, LA = NBUF - 1;           // (DMA) tiles of look-ahead
    constexpr int SMEM = NBUF * BUF > BQW * QG * 32 * FB_OPITCH * 4 ? NBUF * BUF : BQW * QG * 32 * FB_OPITCH * 4;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int tile_id = xcd_remap(blockIdx.x, n_tiles);
    const int4 t = tiles[tile_id];
    const int row_base = t.x, n_tok = t.y, q0 = t.z, head = t.w;
    const int n_kv_tiles = (n_tok + FB_KV - 1) / FB_KV;
    // split mode: sp.krange[tile] = {first 32-key tile, end 32-key tile, part, -} (units of 32 keys, see engine_plan.hip)
    const int4 kr = sp.parts > 1 ? sp.krange[tile_id] : make_int4(0, 2 * n_kv_tiles, 0, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const size_t col0 = (size_t)head * FB_D;

    const bool wave_active = q0 + wave * QG * 32 < n_tok;     // (QG = 2: a wave whose second group lies past the scene computes it on clamped rows)
    // ---- this lane's query (one per group): d = 16 ks + 8 hi + e, pre-scaled, split into bf16 hi / lo ----
    bf16x8 qhg[QG][NKS], ql[NKS];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        int qrow = q0 + (wave * QG + g) * 32 + li;
        if (qrow >= n_tok) qrow = n_tok - 1;      // clamped rows are computed but never stored
        bf16x8 (&qh)[NKS] = qhg[g];
        const float* qrowp = Q + (size_t)(row_base + qrow) * ldq;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            f32x4 x0 = load4<IO>(qrowp, col0 + 8 * hi + 16 * ks), x1 = load4<IO>(qrowp, col0 + 8 * hi + 16 * ks + 4);
            if (IO == 0) {
                x0 *= scale_log2e;
                x1 *= scale_log2e;
            }
            bf16x4 h0, l0, h1, l1;
            planes_of<IO>(x0, h0, l0);
            planes_of<IO>(x1, h1, l1);
            qh[ks] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            ql[ks] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }

    f32x16 og[QG][NO];
#pragma unroll
    for (int g = 0; g < QG; ++g)
#pragma unroll
    for (int b = 0; b < NO; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) og[g][b][r] = 0.f;
    float m_rung[QG], l_rung[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) { m_rung[g] = -INFINITY; l_rung[g] = 0.f; }
    // (tried in round 4 and dropped: the row sums of P through the matrix pipe -- one more MFMA per 16 keys with an all-ones A
    //  operand instead of 32 fp32 adds per lane and tile: 735 vs 790 TFLOP/s on the same box, profiles/r04_probes/flash_bf16_dma_ab.txt)

    // ---- staging: K rows (tid>>4) + 16 i, four d per thread; V: wave-instruction = 4 keys x all 64 d, so that a
    //      16-lane write group fills 4 consecutive 32-byte rows of ONE sub-tile (conflict-free ds_write_b64) ----
    //      (FB_D / 4 threads per K row; per thread FB_D / 16 = NKS pieces of K and of V.  V at other head dims: 32 -> a wave
    //      instruction = 8 keys x 32 d (lane groups 0, 1 = sub-tiles of keys 0..3, groups 2, 3 = of keys 4..7); 128 -> two
    //      instructions per 4 keys, sub-tiles 0..3 and 4..7)
    constexpr int TPR = FB_D / 4, RP = 256 / TPR;
    const int krow = tid / TPR, kc4 = (tid % TPR) * 4;
    const int vg = lane >> 4;
    auto vsub_of = [&](int i) { return FB_D == 32 ? (vg & 1) : FB_D == 64 ? vg : vg + 4 * (i & 1); };
    auto vkey_of = [&](int i) {
        return FB_D == 32 ? 8 * wave + 4 * (vg >> 1) + ((lane >> 2) & 3) + 32 * i
             : FB_D == 64 ? 4 * wave + ((lane >> 2) & 3) + 16 * i
                          : 4 * wave + ((lane >> 2) & 3) + 16 * (i >> 1);
    };
    f32x4 rk[NKS], rv[NKS];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
            int r = kv0 + krow + RP * i;
            r = r < n_tok ? r : n_tok - 1;
            rk[i] = load4<IO>(K + (size_t)(row_base + r) * ldkv, col0 + kc4);
            int rv_ = kv0 + vkey_of(i);
            rv_ = rv_ < n_tok ? rv_ : n_tok - 1;
            rv[i] = load4<IO>(V + (size_t)(row_base + rv_) * ldkv, col0 + vsub_of(i) * 16 + (lane & 3) * 4);
        }
    };
    auto store_tile = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < NKS; ++i) {
            bf16x4 h, l;
            planes_of<IO>(rk[i], h, l);
            char* kp = buf + (krow + RP * i) * FB_KPITCH + kc4 * 2;
            *reinterpret_cast<bf16x4*>(kp) = h;
            if (PL == 2) *reinterpret_cast<bf16x4*>(kp + FB_KPLANE) = l;
            planes_of<IO>(rv[i], h, l);
            char* vp = buf + PL * FB_KPLANE + vsub_of(i) * FB_VSUB + vkey_of(i) * 32 + (lane & 3) * 8;
            *reinterpret_cast<bf16x4*>(vp) = h;
            if (PL == 2) *reinterpret_cast<bf16x4*>(vp + FB_VPLANE) = l;
        }
    };

    // key range of this block in 64-key tiles (the split table counts 32-key tiles; a part boundary inside a 64-key
    // tile is handled by masking, below)
    const int kt0 = kr.x >> 1, kt1 = (kr.y + 1) >> 1;
    const int key_lo = kr.x * 32, key_hi = kr.y * 32 < n_tok ? kr.y * 32 : n_tok;       // keys [key_lo, key_hi) belong to this block
    // ---- LDS-direct staging (DMA): one tile = LPT loads per wave -- K rows in instructions of RPI rows (a wave takes KI consecutive
    //      ones), V in instructions of 32 keys x 32 B of one [64 keys][16 d] sub-tile (a wave takes VI consecutive halves).  The buffer
    //      descriptors span THIS scene's rows only, so keys past the scene's last token read as zeros (their scores are masked below) ----
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K + (size_t)row_base * ldkv), 0, DMA ? (int)(unsigned)((size_t)n_tok * ldkv * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V + (size_t)row_base * ldkv), 0, DMA ? (int)(unsigned)((size_t)n_tok * ldkv * 4) : 0, 0x00020000);
    const unsigned ld4 = (unsigned)ldkv * 4u;
    auto kswz_of = [](int row) { return FB_D == 32 ? (row >> 2) & 3 : FB_D == 64 ? (row >> 1) & 7 : row & 15; };
    unsigned vK[4];                                   // (KI <= 4 used; an array bound that depends on the template arguments makes hipcc drop every host stub of this template without a diagnostic)
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const int row = (wave_u * KI + j) * RPI + lane / CPR;
        vK[j] = (unsigned)row * ld4 + (unsigned)col0 * 2u + (unsigned)(((lane % CPR) ^ kswz_of(row)) << 4);
    }
    const unsigned vV0 = (unsigned)(lane >> 1) * ld4 + (unsigned)col0 * 2u + (unsigned)(lane & 1) * 16u;       // key lane / 2 of a 32-key half, 16-byte half lane & 1
    auto dma_tile = [&](int kv0, char* buf) {
        const unsigned s0 = (unsigned)kv0 * ld4;
#pragma unroll
        for (int j = 0; j < KI; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, buf + (wave_u * KI + j) * 1024, 16, vK[j], s0, 0, 0);
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int idx = wave_u * VI + j, sub = idx >> 1, kh = idx & 1;                 // sub-tile, 32-key half of it
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, buf + KBYTES + sub * FB_VSUB + kh * 1024, 16, vV0, s0 + (unsigned)kh * 32u * ld4 + (unsigned)sub * 32u, 0, 0);
        }
    };
    // counted wait: everything but the newest `n` tiles (LPT loads each) has landed
    auto wait_tiles = [&](int n) {
        if (n <= 0) fb_wait_vm<0>();
        else if (n == 1) fb_wait_vm<LPT>();
        else if (n == 2) fb_wait_vm<2 * LPT>();
        else fb_wait_vm<3 * LPT>();
    };
    if (DMA) {
        int issued = 0;
#pragma unroll
        for (int j = 0; j < LA; ++j)
            if (kt0 + j < kt1) { dma_tile((kt0 + j) * FB_KV, smem + j * BUF); ++issued; }
        wait_tiles(issued - 1);
        asm volatile("s_barrier" ::: "memory");
    } else {
        if (kt0 < kt1) {
            load_tile(kt0 * FB_KV);
            store_tile(smem + (kt0 & 1) * BUF);
        }
        __syncthreads();
    }

    // One key tile.  SLOT: the tile's LDS buffer as a compile-time constant where the loop below can provide it (two-buffer
    // configurations: the loop is unrolled by two, so every LDS address of the iteration is lane constant + immediate -- the
    // run-time slot cost ~40 VALU address instructions per tile in a kernel that is VALU-bound, round 5), else -1 = `ring`.
    int ring = 0;                                          // (DMA) buffer of tile kt
    auto tile_step = [&](int kt, auto slotc) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slotc)::value;
        const int buf_i = SLOT >= 0 ? SLOT : (DMA ? ring : (kt & 1));
        const char* sK = smem + buf_i * BUF;
        const char* sV = sK + (DMA ? KBYTES : PL * FB_KPLANE);
        const bool more = kt + 1 < kt1;
        if (DMA) {
            // tile kt + LA goes into the buffer tile kt - 1 was read from: every wave passed the barrier that ended iteration kt - 1
            if (kt + LA < kt1 && !(kExperiments && (sp.ablate & 1))) dma_tile((kt + LA) * FB_KV, smem + (buf_i == 0 ? NBUF - 1 : buf_i - 1) * BUF);
        } else if (more && !(kExperiments && (sp.ablate & 1))) load_tile((kt + 1) * FB_KV);

        if (wave_active) {
            // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d], two blocks of 32 keys (per query group) ----
            f32x16 sg[QG][2];
#pragma unroll
            for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sg[g][0][r] = 0.f; sg[g][1][r] = 0.f; }
            {
                // the two 32-key blocks alternate, so consecutive MFMAs never wait for each other's accumulator
                const char* kp = sK + li * FB_KPITCH + 16 * hi;
                const char* kd = sK + li * ROWB;                      // (DMA) row li; rows li and li + 32 share the swizzle
                const int kswz = kswz_of(li);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const bf16x8 kh0 = *reinterpret_cast<const bf16x8*>(DMA ? kd + (((hi + 2 * ks) ^ kswz) << 4) : kp + 32 * ks);
                    const bf16x8 kh1 = *reinterpret_cast<const bf16x8*>(DMA ? kd + 32 * ROWB + (((hi + 2 * ks) ^ kswz) << 4) : kp + 32 * FB_KPITCH + 32 * ks);
                    if (PL == 2) {
                        const bf16x8 kl0 = *reinterpret_cast<const bf16x8*>(kp + FB_KPLANE + 32 * ks);
                        const bf16x8 kl1 = *reinterpret_cast<const bf16x8*>(kp + FB_KPLANE + 32 * FB_KPITCH + 32 * ks);
                        sg[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl0, qhg[0][ks], sg[0][0], 0, 0, 0);
                        sg[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl1, qhg[0][ks], sg[0][1], 0, 0, 0);
                        sg[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh0, ql[ks], sg[0][0], 0, 0, 0);
                        sg[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh1, ql[ks], sg[0][1], 0, 0, 0);
                    }
                    if ((ABL & 64)) {           // no QK MFMAs (the fragments stay read)
                        asm volatile("" :: "v"(kh0), "v"(kh1));
                        continue;
                    }
#pragma unroll
                    for (int g = 0; g < QG; ++g) {
                        sg[g][0] = mfma_h<F16>(kh0, qhg[g][ks], sg[g][0]);
                        sg[g][1] = mfma_h<F16>(kh1, qhg[g][ks], sg[g][1]);
                    }
                }
            }
            // The first readers of the score registers below are inline-asm VALU instructions, which hipcc's hazard recogniser
            // does not see: the wait states between an MFMA's register write and a VALU read of it (19 for a 16-pass MFMA; the
            // hardware has no interlock there) are inserted by hand.  The asm names the accumulators, so it cannot move above the
            // MFMAs, and everything that reads them is ordered behind it.  (Found by the kernel tests: without it the maxima were
            // taken from stale registers.)
            if (QG == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sg[0][0]), "+v"(sg[0][1]), "+v"(sg[QG - 1][0]), "+v"(sg[QG - 1][1]));
            else asm volatile("s_nop 15\n\ts_nop 3" : "+v"(sg[0][0]), "+v"(sg[0][1]));
            const int kv0 = kt * FB_KV;
            // ---- online softmax for this lane's query ----
            // The kernel is bound by VALU issue at head dim 64 (round 4 PMC: matrix pipe 35 % busy; per tile and wave 512 MFMA cycles
            // against ~1100 of VALU), so the softmax is written for instruction count (round 5): the maximum of the 32 scores with
            // v_max3_f32 (two new values per instruction; fmaxf costs a canonicalising v_max per MFMA result on top of the max itself:
            // 54 -> 16 instructions), score - m and the row sum as packed fp32 pairs (v_pk_add_f32: 33 + 33 -> 16 + 16); the 32
            // quarter-rate v_exp_f32 stay.
            auto softmax_group = [&](f32x16 (&s)[2], f32x16 (&o)[NO], float& m_run, float& l_run) __attribute__((always_inline)) {
            // keys outside [key_lo, key_hi): beyond the scene's tokens (last tile) or another part's (split mode)
            if (kv0 < key_lo || kv0 + FB_KV > key_hi) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + 32 * kb + crow32(r, hi);
                        if (key < key_lo || key >= key_hi) s[kb][r] = -INFINITY;
                    }
            }
            float mx;
            {
                float ma = max3f(s[0][0], s[0][1], s[0][2]), mb = max3f(s[1][0], s[1][1], s[1][2]);    // two chains: no max waits for its predecessor
#pragma unroll
                for (int r = 3; r < 15; r += 2) { ma = max3f(ma, s[0][r], s[0][r + 1]); mb = max3f(mb, s[1][r], s[1][r + 1]); }
                mx = max3f(ma, mb, s[0][15]);
                mx = max3f(mx, s[1][15], s[1][15]);
            }
            if (!((ABL & 16))) { float ha, hb; half_swap(mx, ha, hb); mx = max3f(ha, hb, hb); }       // (ablate 16: no cross-half exchanges)
            if ((ABL & 8)) mx = 0.f;                                        // (ablate 8: no maximum at all)
            const float m_new = max3f(m_run, mx, mx);
            // (a part whose first tile is fully masked for this query keeps m = -inf; exp2(-inf - -inf) must not be NaN)
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            f32x2 rs2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 x = pk_sub(f32x2{s[kb][r], s[kb][r + 1]}, m_use);
                    if (!((ABL & 4))) {                                      // (ablate 4: no exponentials)
                        x[0] = __builtin_amdgcn_exp2f(x[0]);
                        x[1] = __builtin_amdgcn_exp2f(x[1]);
                    }
                    rs2 = pk_add(rs2, x);
                    s[kb][r] = x[0];
                    s[kb][r + 1] = x[1];
                }
            float rs = rs2[0] + rs2[1];
            if (!((ABL & 16))) rs = half_sum(rs);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int b = 0; b < NO; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            }
            };
            // ---- O^T[d][query] += sum_key V[key][d] * P[key][query]: groups [G0, G1) share every V fragment ----
            auto pv_groups = [&](auto g0c, auto g1c) __attribute__((always_inline)) {
            constexpr int G0 = decltype(g0c)::value, G1 = decltype(g1c)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kb = j >> 1, half = j & 1;
                bf16x8 phg[QG], pl;
#pragma unroll
                for (int g = G0; g < G1; ++g) {
                f32x4 p0, p1;
#pragma unroll
                for (int c = 0; c < 4; ++c) { p0[c] = sg[g][kb][8 * half + c]; p1[c] = sg[g][kb][8 * half + 4 + c]; }
                bf16x4 h0, l0, h1, l1;
                if constexpr (F16) {              // probabilities in [0, 1]: fp16, no clamp needed
                    typedef _Float16 f16x4_p __attribute__((ext_vector_type(4)));
                    h0 = l0 = __builtin_bit_cast(bf16x4, __builtin_convertvector(p0, f16x4_p));
                    h1 = l1 = __builtin_bit_cast(bf16x4, __builtin_convertvector(p1, f16x4_p));
                } else {
                    split4(p0, h0, l0);
                    split4(p1, h1, l1);
                }
                phg[g] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                const bf16x8 ph = phg[G0];
                f32x16 (&o)[NO] = og[G0];
                const int k0 = 32 * kb + 16 * half + 4 * hi;          // first key of this lane half's k-slots (then +8)
                bf16x8 vf[NO][PL];
#pragma unroll
                for (int db = 0; db < NO; ++db)
#pragma unroll
                    for (int pln = 0; pln < PL; ++pln) {
                        const char* vb = sV + pln * FB_VPLANE + (2 * db + ((lane >> 4) & 1)) * FB_VSUB;
                        if (TR) {
                            const char* a = vb + (k0 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
                            const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3)))*)(a));
                            const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (s16x4 __attribute__((address_space(3)))*)(a + 8 * 32));
                            const bf16x4 b0 = __builtin_bit_cast(bf16x4, x0), b1 = __builtin_bit_cast(bf16x4, x1);
                            vf[db][pln] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                        } else {
                            const unsigned short* g = reinterpret_cast<const unsigned short*>(vb) + (lane & 15);
                            typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
                            u16x8 u;
#pragma unroll
                            for (int e = 0; e < 8; ++e) u[e] = g[(k0 + 8 * (e >> 2) + (e & 3)) * 16];
                            vf[db][pln] = __builtin_bit_cast(bf16x8, u);
                        }
                    }
                // the d-blocks alternate (independent accumulators back to back)
                if (PL == 2) {
#pragma unroll
                    for (int db = 0; db < NO; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][PL - 1], ph, o[db], 0, 0, 0);
                    if (PVT == 3) {
#pragma unroll
                        for (int db = 0; db < NO; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db][0], pl, o[db], 0, 0, 0);
                    }
                }
                if ((ABL & 32)) {               // no PV MFMAs (P converted, V fragments read)
                    asm volatile("" :: "v"(ph), "v"(vf[0][0]), "v"(vf[NO - 1][0]));
                    continue;
                }
#pragma unroll
                for (int g = G0; g < G1; ++g)
#pragma unroll
                for (int db = 0; db < NO; ++db) og[g][db] = mfma_h<F16>(vf[db][0], phg[g], og[g][db]);
            }
            };
            typedef std::integral_constant<int, 0> I0;
            typedef std::integral_constant<int, 1> I1;
            typedef std::integral_constant<int, QG> IQ;
            if constexpr (QG == 2 && ORD == 1) {
                softmax_group(sg[0], og[0], m_rung[0], l_rung[0]);
                pv_groups(I0{}, I1{});
                softmax_group(sg[QG - 1], og[QG - 1], m_rung[QG - 1], l_rung[QG - 1]);
                pv_groups(I1{}, IQ{});
            } else {
#pragma unroll
                for (int g = 0; g < QG; ++g) softmax_group(sg[g], og[g], m_rung[g], l_rung[g]);
                pv_groups(I0{}, IQ{});
            }
        }   // wave_active
        if (DMA) {
            // tile kt + 1 has landed once only the tiles issued after it (kt + 2 .. kt + LA, as far as they exist) are outstanding
            wait_tiles((kExperiments && (sp.ablate & 1)) ? 0 : min(LA - 1, kt1 - kt - 2));
            if ((ABL & 256)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (ablate 256: no barrier per tile)
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            ring = ring == NBUF - 1 ? 0 : ring + 1;
        } else {
            if (more && !(kExperiments && (sp.ablate & 2))) store_tile(smem + ((kt + 1) & 1) * BUF);
            __syncthreads();
        }
    };
    if (DMA && NBUF == 2) {                                 // the first tile of the range sits in buffer 0
        for (int kt = kt0; kt < kt1; kt += 2) {
            tile_step(kt, std::integral_constant<int, 0>{});
            if (kt + 1 < kt1) tile_step(kt + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int kt = kt0; kt < kt1; ++kt) tile_step(kt, std::integral_constant<int, -1>{});
    }

    // ---- normalise (or, in split mode, keep un-normalised and record m, l), transpose through LDS
    //      (wave-private [32 q][68]), coalesced store ----
    const bool split = sp.parts > 1;
    if (split) O = sp.o_part + (size_t)kr.z * sp.part_stride;
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const float m_run = m_rung[g], l_run = l_rung[g];
        const float inv_l = split ? 1.f : 1.f / l_run;
        if (split) {
            const int qr = q0 + (wave * QG + g) * 32 + li;
            if (hi == 0 && qr < n_tok) {
                const size_t i = ((size_t)kr.z * sp.rows + row_base + qr) * sp.heads + head;
                sp.m_part[i] = m_run;
                sp.l_part[i] = l_run;
            }
        }
        float* so = reinterpret_cast<float*>(smem) + (wave * QG + g) * (32 * FB_OPITCH);
#pragma unroll
        for (int b = 0; b < NO; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = og[g][b][r] * inv_l;
                so[li * FB_OPITCH + 32 * b + crow32(r, hi)] = (IO == 1 && !split) ? pack_split(a) : a;   // (split-key partials stay fp32: the merge packs)
            }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < QG; ++g) {
    const float* so = reinterpret_cast<const float*>(smem) + (wave * QG + g) * (32 * FB_OPITCH);
#pragma unroll
    for (int i = 0; i < FB_D / 8; ++i) {
        const int idx = lane + 64 * i;             // 8 FB_D float4 = 32 rows x FB_D / 4
        const int r = idx / (FB_D / 4), c4 = (idx % (FB_D / 4)) * 4;
        const int qr = q0 + (wave * QG + g) * 32 + r;
        if (qr < n_tok) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(so + r * FB_OPITCH + c4);
            float* orow = O + (size_t)(row_base + qr) * ldo;
            if (IO == 3 && !split) {
                typedef _Float16 f16x4_o __attribute__((ext_vector_type(4)));
                f32x4 vc;
#pragma unroll
                for (int c = 0; c < 4; ++c) vc[c] = __builtin_amdgcn_fmed3f(v[c], -65504.f, 65504.f);
                *reinterpret_cast<f16x4_o*>(reinterpret_cast<char*>(orow) + (col0 + c4) * 2) = __builtin_convertvector(vc, f16x4_o);
            } else if (IO == 2 && !split)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(orow) + (col0 + c4) * 2) = __builtin_convertvector(v, bf16x4);
            else
                *reinterpret_cast<f32x4*>(orow + col0 + c4) = v;
        }
    }
    }
}


}  // namespace

// head dims other than 64 exist for the tensor formats of the bf16 modes only (io_split 1 | 2, transpose read), and
// split-bf16 (two LDS planes) not at 128, where the tile buffers of two blocks no longer fit a CU
bool flash_attn_bf16_supports(int head_dim, int terms, int use_tr, int io_split) {
    if (head_dim == 64) return true;
    if (head_dim != 32 && head_dim != 128) return false;
    if (!use_tr || !io_split) return false;
    if (io_split >= 2 && terms != 1) return false;
    return head_dim == 32 || terms == 1;
}

int launch_flash_attn_bf16(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo,
                           const int4* tiles, int n_tiles, float scale_log2e, int terms, int use_tr, int io_split,
                           hipStream_t s, const FlashSplit* split, int pv_terms, int head_dim) {
    if (n_tiles <= 0) return 0;
    const int FB_D = head_dim;
    if (!flash_attn_bf16_supports(head_dim, terms, use_tr, io_split)) return fail(-1, "flash_attn_bf16: head dim / format combination not built");
    if ((ldq | ldkv | ldo) & 3) return fail(-1, "flash_attn: leading dims must be multiples of 4");
    if (terms != 1 && terms != 3) return fail(-1, "flash_attn_bf16: terms must be 1 or 3");
    if (io_split == 3 && !(split && split->rows > 0 && (size_t)split->rows * (size_t)ldkv * 4 < (1ull << 32)))
        return fail(-1, "flash_attn_bf16: fp16 half rows are built for scenes addressable with 32-bit offsets (the LDS-direct kernel)");
    FlashSplit sp{};
    if (split && split->parts > 1) {
        sp = *split;
        if (!sp.krange || !sp.o_part || !sp.m_part || !sp.l_part || sp.heads * FB_D > ldo)
            return fail(-1, "flash_attn: incomplete split-key workspace");
    }
    if (split) { sp.ablate = split->ablate; sp.bq = split->bq; sp.rows = split->rows; sp.qg = split->qg; }
    if (sp.bq != FLASH_BQ && !(sp.bq == FLASH_BQ_BIG && FB_D == 64 && io_split >= 2 && use_tr == 1 && sp.parts <= 1 && sp.rows > 0 &&
                               (size_t)sp.rows * (size_t)ldkv * 4 < (1ull << 32)))
        return fail(-1, "flash_attn_bf16: 256-query tiles are built for half rows, head dim 64, the LDS-direct kernel, no key split");
    // The LDS-direct K / V staging addresses a scene with 32-bit byte offsets (buffer descriptor of n_tok * ldkv * 4 bytes, row * ld4
    // VGPR offsets): only where every scene's rows are known to span less than 4 GiB -- the caller states the rows of the whole
    // tensor in FlashSplit::rows (an upper bound of any scene, and of the batch-wide attention of batch_mode 'reference').  Unknown
    // or larger: the register-staged kernel, which addresses rows with size_t.
    if (io_split == 2 && use_tr && use_tr != 2 && !(split && split->rows > 0 && (size_t)split->rows * (size_t)ldkv * 4 < (1ull << 32))) use_tr = 2;
#define VLSAT_FA(T, R, S) hipLaunchKernelGGL((flash_attn_bf16_kernel<T, R, S>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp)
#define VLSAT_FAD(T, S, P, D) hipLaunchKernelGGL((flash_attn_bf16_kernel<T, true, S, P, D>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp)
    if (FB_D != 64) {          // 16 / 4 heads: the formats the forward uses (the transpose-read path; split-bf16 only at 32)
        if (io_split == 3) {                         // fp16 half rows
            if (use_tr == 2 || !use_tr || terms != 1) return fail(-1, "flash_attn_bf16: fp16 half rows are built for the LDS-direct single-rounding kernel only");
            if (FB_D == 32) hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 3, 3, 32, 2>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
            else hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 3, 3, 128, 2>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        } else
        if (io_split == 2 && use_tr != 2) {          // half rows: LDS-direct K/V staging, one tile ahead
            if (FB_D == 32) hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 32, 2>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
            else hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 128, 2>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        } else if (FB_D == 32) {
            if (io_split == 2) VLSAT_FAD(1, 2, 3, 32);
            else if (terms == 3 && pv_terms == 2) VLSAT_FAD(3, 1, 2, 32);
            else if (terms == 3) VLSAT_FAD(3, 1, 3, 32);
            else VLSAT_FAD(1, 1, 3, 32);
        } else {
            if (io_split == 2) VLSAT_FAD(1, 2, 3, 128); else VLSAT_FAD(1, 1, 3, 128);
        }
    } else
    if (io_split == 3) {                 // fp16 half rows (precision mode fp16_mixed): the two shipped forms of the LDS-direct kernel
        if (!use_tr || terms != 1 || use_tr == 2 || use_tr >= 3) return fail(-1, "flash_attn_bf16: fp16 half rows are built for the LDS-direct single-rounding kernel only");
        if (sp.bq == FLASH_BQ_BIG)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 3, 3, 64, 2, 8>), dim3(n_tiles), dim3(512), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 3, 3, 64, 2>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
    } else
    if (io_split == 2) {
        if (!use_tr || terms != 1) return fail(-1, "flash_attn_bf16: half-row tensors need terms = 1 and the transpose-read path");
        // LDS-direct K/V staging (scene-relative 32-bit byte offsets: checked above).  Measured on one box, interleaved (profiles/r04_probes/flash_bf16_dma_ab.txt): register-
        // staged 596 TFLOP/s; ring of 3 buffers (3 blocks per CU) 745-773; ring of 4 (2 blocks per CU) 650-659; ring of 2 = one
        // tile ahead with FOUR blocks per CU (34 KB of LDS, 120 VGPRs) 813-817: occupancy beats look-ahead depth.
        if (use_tr == 3)            // (experiments: vlsat_debug_option "flash_dma" 3 | 4 = rings of three / four buffers)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 3>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else if (use_tr == 4)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 4>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
#ifdef VLSAT_EXPERIMENTS
#define VLSAT_FA_ABL(A) else if (use_tr != 2 && sp.bq == FLASH_BQ_BIG && (sp.ablate & ~3) == (A)) \
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 2, 8, A>), dim3(n_tiles), dim3(512), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        VLSAT_FA_ABL(4) VLSAT_FA_ABL(8) VLSAT_FA_ABL(16) VLSAT_FA_ABL(28) VLSAT_FA_ABL(32) VLSAT_FA_ABL(64) VLSAT_FA_ABL(96) VLSAT_FA_ABL(256) VLSAT_FA_ABL(124) VLSAT_FA_ABL(380)
#undef VLSAT_FA_ABL
#endif
        else if (use_tr != 2 && sp.bq == FLASH_BQ_BIG && sp.qg == 1)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 2, 4, 0, 2, 0>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else if (use_tr != 2 && sp.bq == FLASH_BQ_BIG && sp.qg == 2)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 2, 4, 0, 2, 1>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else if (use_tr != 2 && sp.qg == 1 && sp.parts <= 1)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 2, 2, 0, 2, 0>), dim3(n_tiles), dim3(128), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else if (use_tr != 2 && sp.qg == 2 && sp.parts <= 1)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 2, 2, 0, 2, 1>), dim3(n_tiles), dim3(128), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else if (use_tr != 2 && sp.bq == FLASH_BQ_BIG)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 2, 8>), dim3(n_tiles), dim3(512), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else if (use_tr != 2)       // (use_tr = 2: the register-staged kernel of round 3, for A/B -- "flash_dma" 0)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<1, true, 2, 3, 64, 2>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else
            VLSAT_FA(1, true, 2);
    } else if (io_split) {
        if (!use_tr) return fail(-1, "flash_attn_bf16: the split-pair format is built for the transpose-read path only");
        if (terms == 3 && pv_terms == 2)
            hipLaunchKernelGGL((flash_attn_bf16_kernel<3, true, 1, 2>), dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
        else if (terms == 3) VLSAT_FA(3, true, 1); else VLSAT_FA(1, true, 1);
    } else if (terms == 3) { if (use_tr) VLSAT_FA(3, true, 0); else VLSAT_FA(3, false, 0); }
    else                   { if (use_tr) VLSAT_FA(1, true, 0); else VLSAT_FA(1, false, 0); }
#undef VLSAT_FA
#undef VLSAT_FAD
    VLSAT_LAUNCH_CHECK("flash_attn_bf16");
    if (sp.parts > 1) return launch_flash_merge(O, ldo, sp, s, io_split, FB_D);
    return 0;
}

}  // namespace vlsat
