// bf16 GEMM of the single-rounding modes for the large edge-row launches (BASELINE configs[2]) -- same contract as
// gemm_f32.hip / gemm_bf16_ring.hip,
//     C[M,N] = act((A[M,K] . W[N,K]^T) + bias + resid_scale*resid + g0[gi0] + g1[gi1]) * c_scale,
// for A stored as HALF ROWS (bf16 at byte 2 k of an fp32-pitched row) and one bf16 weight plane.  Every nn.Linear on edge
// rows goes through it in the bf16_mixed / bf16 modes (reference network_MMG.py:59-79,93-98; attention.py:54-58,77;
// network_PointNet.py:328-341).
//
// Structure (cdna_hip_programming.md "The 256^2 8-phase template", rebuilt for this library's operand formats):
//   * ONE persistent block of 8 waves per CU, block tile 256 x 256, waves 2 (M) x 4 (N), wave tile 128 x 64 =
//     4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (transposed product: a lane owns one output ROW, gemm_core.h).
//   * K in tiles of 64; LDS holds two K-tiles (2 x 64 KB), each as four HALF-TILES of 128 rows x 128 bytes:
//       A_h = rows {wr*128 + h*64 + i} of the block's A panel,  W_h = rows {wc*64 + h*32 + j} of its weight panel,
//     i.e. half-tile h is what quadrant h of EVERY wave reads.  Rows are 128 bytes, 16-byte chunks XOR-swizzled with
//     (row >> 1) & 7 on the global side of the LDS-direct loads and on the fragment reads (0 bank conflicts in the K loop;
//     the epilogue's transposition buffer has its own swizzle, see epi_t).
//   * A K-tile is FOUR PHASES, one per quadrant of the wave tile, in the order (a0,w0) (a0,w1) (a1,w1) (a1,w0); a phase is
//         ds_read the operands the quadrant does not hold yet (12 / 4 / 8 / 0 ds_read_b128)
//         issue ONE half-tile of LDS-direct loads (2 x buffer_load_dwordx4 ... lds per lane)
//         s_waitcnt vmcnt(8 + e) ; s_barrier ; s_waitcnt lgkmcnt(0) ; 8 MFMAs under s_setprio 1 ; s_barrier
//     and the two wave rows run one barrier apart, so on every SIMD one wave is in its MFMA cluster while its partner
//     reads LDS and issues loads.
//   * Loads are never drained: phase q of K-tile g stages  W_1(g+1), A_1(g+1), A_0(g+2), W_0(g+2)  for q = 1..4 -- each
//     into the buffer half its previous tenant's last ds_read left >= 2 phases earlier -- and every wait is counted:
//     "everything but the four half-tiles issued last (and the e epilogue stores issued since) has landed", which is
//     exactly what the next phase reads.  The sequence of K-tiles runs on across output tiles (the staging cursor is two
//     K-tiles ahead, in the next tile if need be).
//   * The epilogue is spread over the phases around a tile boundary, one 32-row strip of the wave tile per phase: strips
//     0 / 1 (final after quadrant (a0,w1)) in phases 3 / 4 of the tile's last K-tile, strips 2 / 3 in phases 1 / 2 of the
//     NEXT tile's first K-tile -- under the partner wave's MFMAs.  A strip goes bias -> activation -> bf16 -> a wave-private
//     4 KB LDS transposition -> buffer_store_dwordx4 of 8 full 128-byte rows per instruction (row-per-lane 8-byte stores
//     touch 32 lines per instruction and made the epilogue 40 % of the launch).
// Results equal the 128 x 128 kernel's up to the place of the bias in the fp32 summation (first instead of last).
// Roofline: bf16 MFMA (2.5 PF dense); per K-tile a block moves 64 KB from L2 for 8.4 MFLOP.
#include <type_traits>

#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

template <int N> __device__ __forceinline__ void p8_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void p8_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void p8_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 64;
constexpr int P8_HALF = 128 * 128;            // bytes of a half-tile
constexpr int P8_WBASE = 4 * P8_HALF;         // weight half-tiles start after the four A half-tiles (2 buffers x 2 halves)
constexpr int P8_TBASE = 8 * P8_HALF;         // wave-private transposition buffers of the epilogue (8 x 4 KB)

// (ABL bit 4: no barrier after the MFMA clusters, bit 5: no barrier in the K loop at all -- timing experiments)
enum { P8_MID = 0, P8_LAST = 1, P8_FIRST = 2, P8_SECOND = 3, P8_FIRST0 = 4 };   // FIRST0: first K-tile of a tile without strips of a previous one

// CF: storage format of C (0 fp32, 2 half rows of bf16, 3 half rows of FP16 clamped to +-65504 -- the out-projection of the single-rounded edge
// attention, whose only reader is the LayerNorm: round 6).  ABL (timing experiments, results are garbage): bit 0 no LDS-direct
// loads after the prologue, bit 1 no MFMAs, bit 2 no fragment reads, bit 3 no epilogue stores
// F32: exact-fp32 operands (A fp32 rows, W fp32 [N,K], v_mfma_f32_32x32x2_f32, K-tiles of 32): the same LDS image -- 128-byte
// rows, a lane's fragment is the 16-byte chunk 2 ks + hi = four consecutive k (gemm_core.h, K-permutation trick) -- so
// loader, phases, waits and epilogue are shared; a phase is 32 MFMAs of 64 cycles there.  Bias then joins after the
// transposition (8 registers per lane; same fp32 operation order as gemm_f32.hip: bit-identical results).
// MODE 2 (X3): split-bf16, three MFMAs per product.  A = split-pair words (bf16 hi | lo in one 32-bit word, common.h), the
// same 128-byte rows as fp32 (K-tiles of 32), separated with v_perm on the fragment side; a weight half-tile is the hi and
// the lo plane side by side, 128 rows x 64 bytes each (chunks swizzled with (row >> 2) & 3), one LDS-direct load per plane
// -- so a half-tile is still two loads and every counted wait is unchanged.  Term order per accumulator and epilogue order
// as in the older split-bf16 kernels: bit-identical results.
// (round 6, measured and dropped: the g0 rows of a gathered-row tile DEDUPLICATED per 32-row strip -- source-major edge lists name
//  one or two g0 rows per strip; each distinct 256-byte segment fetched once by 16 lanes into the wave's idle transposition
//  buffer, every lane reading its run's values from LDS, bit-identical results -- 213.6 / 218.5 us against 209.5 / 210.6 us for the
//  per-lane gather on the bench batch's indices, 9852-9867 vs 9839-9880 scenes/s per step: the repeated g0 lines were L1 hits
//  already, what the launch pays for is the L2 -> CU traffic of the DISTINCT lines, which the deduplication does not change.
//  profiles/r06_probes/p8_dedup_krot_{kernel,step}_ab.txt)
template <int MODE, int ADD, bool RELU, int CF, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(GemmArgs p, int n_tiles, int nbn) {
    constexpr bool F32 = MODE == 1, X3 = MODE == 2, H16 = MODE == 3, WORDS = F32 || X3;      // WORDS: 4-byte A elements, K-tiles of 32; H16: MODE 0 on fp16 (GemmArgs::half_f16)
    static_assert(MODE == 3 ? (CF == 0 || CF == 3) : MODE == 0 ? (CF == 0 || CF == 2 || CF == 3) : F32 ? CF == 0 : (CF == 0 || CF == 1), "output format of the mode");
    __shared__ __attribute__((aligned(16))) char smem[10 * P8_HALF];
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr bool SPREAD = ADD == 0;             // epilogue strips 2 / 3 inside the next tile's first K-tile
    // (ABL bits 8, 9, gathered-row launches: 256 = the init loads in a row-contiguous lane pattern, 512 = no init loads)
    constexpr bool HALF = CF >= 2;                // two bytes per element of C
    constexpr int E = HALF ? 4 : 8;            // buffer stores per epilogue strip and lane

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 31, hi = lane >> 5;
    const int g8 = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int KT = p.K / (WORDS ? 32 : P8_BK);         // K-tiles of 128 bytes per A row
    auto tile_of_round = [&](int r) { return (r * 8 + xcd) * g8 + slot; };
    if (tile_of_round(0) >= n_tiles) return;

    // ---- LDS-direct loader: lane constants (a wave instruction fills 8 rows x 128 bytes) ----
    const int srow = wave * 8 + (lane >> 3);                                   // LDS row inside a 64-row round
    const unsigned schunk = (unsigned)((lane & 7) ^ ((srow >> 1) & 7));          // logical chunk this lane fetches
    constexpr unsigned EA = WORDS ? 4u : 2u, EW = F32 ? 4u : 2u;              // bytes per A / W element
    constexpr unsigned KA = 128u, KW = X3 ? 64u : 128u;                        // bytes of a K-tile in an A / W row
    const unsigned lda4 = (unsigned)p.lda * 4u, ldw2 = (unsigned)p.ldw * EW;     // row pitches in bytes (half rows keep the fp32 pitch)
    const unsigned vA = (unsigned)srow * lda4 + schunk * 16u;
    // weight rows: 128-byte K-tiles like A (8 rows per wave instruction), or -- X3 -- 64-byte ones (16 rows, one plane)
    const int wrow3 = wave * 16 + (lane >> 2);                                  // X3: LDS row of the half-tile's plane
    const unsigned vW = X3 ? (unsigned)((wrow3 >> 5) * 64 + (wrow3 & 31)) * ldw2 + (unsigned)((lane & 3) ^ ((wrow3 >> 2) & 3)) * 16u
                           : (unsigned)((wave >> 2) * 64 + (wave & 3) * 8 + (lane >> 3)) * ldw2 + schunk * 16u;
    const int na = (int)((size_t)(p.M - 1) * lda4 + (size_t)p.K * EA), nw = (int)(((size_t)(p.N - 1) * p.ldw + p.K) * EW);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, na, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(F32 ? (void*)const_cast<float*>(p.W) : (void*)const_cast<uint16_t*>(p.Whi), 0, nw, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(X3 ? p.Wlo : p.Whi), 0, nw, 0x00020000);
    char* const sdst = smem + wave * 1024;
    // two wave instructions = one half-tile: rows 0..63 and 64..127 of it
    auto ld2 = [&](const __amdgpu_buffer_rsrc_t& r, char* dst, unsigned v, unsigned s0, unsigned step) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, v, s0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst + 8192, 16, v, s0 + step, 0, 0);
    };
    // weight half-tile h at byte offset sw (K-tile included): two 64-row rounds of one plane, or (X3) the two planes
    auto ldw = [&](char* dst, unsigned sw, int h) {
        const unsigned s0 = sw + (unsigned)(h * 32) * ldw2;
        if (X3) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, vW, s0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, dst + 8192, 16, vW, s0, 0, 0);
        } else {
            ld2(rw, dst, vW, s0, 128u * ldw2);
        }
    };
    // half-tile h of the K-tile at byte offsets (sa, sw) into buffer `buf`
    auto stage_a = [&](int buf, int h, unsigned sa) {
        if (!(ABL & 1)) ld2(ra, sdst + (buf * 2 + h) * P8_HALF, vA, sa + (unsigned)(h * 64) * lda4, 128u * lda4);
    };
    auto stage_w = [&](int buf, int h, unsigned sw) {
        if (!(ABL & 1)) ldw(sdst + P8_WBASE + (buf * 2 + h) * P8_HALF, sw, h);
    };

    // ---- staging cursor: the K-tile sequence of this block, across output tiles ----
    // (past the block's last tile the cursor stays on it: the loads keep their rhythm, so every wait is a counted one and
    //  the K loop has no branch; what they fetch is never read)
    struct Cur { int r, kt; unsigned sa, sw; };
    auto locate = [&](Cur& c) {
        const int v = tile_of_round(c.r);
        if (v >= n_tiles) return;
        const int tm = v / nbn, tn = v - tm * nbn;
        c.sa = (unsigned)(tm * P8_BM) * lda4;
        c.sw = (unsigned)(tn * P8_BN) * ldw2;
    };
    auto advance = [&](Cur& c) {
        if (++c.kt == KT) { c.kt = 0; ++c.r; locate(c); }
    };
    // (A/B, GemmArgs::k_rot) column tile tn of a row panel walks its K-tiles starting at tn * k_rot: the blocks that share an A panel
    //  (neighbouring slots of one XCD) then ask L2 for the same lines a K-tile or more apart instead of in the same microsecond
    const int rot = p.k_rot ? ((tile_of_round(0) % nbn) * p.k_rot) % KT : 0;
    auto kti = [&](const Cur& c) { const int j = c.kt + rot; return (unsigned)(j >= KT ? j - KT : j); };
    Cur c1{0, 0, 0u, 0u};
    locate(c1);
    const Cur c0 = c1;
    advance(c1);
    Cur c2 = c1;
    advance(c2);

    // ---- fragment readers: per-lane byte offsets of k-step ks (the chunk XOR is not an add: one register per k-step) ----
    // (X3: fragment i of an A tile is chunk 4 (i >> 1) + 2 hi + (i & 1) -- eight words of k-step i >> 1 in two reads; weight
    //  fragment i is chunk 2 (i & 1) + hi of the 64-byte row of plane i >> 1)
    const int swz = (li >> 1) & 7;
    int offA[4], offW[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int ca = X3 ? (((ks >> 1) << 2) | (hi << 1) | (ks & 1)) : 2 * ks + hi;
        offA[ks] = (wr * 64 + li) * 128 + ((ca ^ swz) * 16);
        offW[ks] = X3 ? P8_WBASE + (ks >> 1) * 8192 + (wc * 32 + li) * 64 + (((2 * (ks & 1) + hi) ^ ((li >> 2) & 3)) * 16)
                      : P8_WBASE + (wc * 32 + li) * 128 + (((2 * ks + hi) ^ swz) * 16);
    }
    bf16x8 af[2][4], w0[4], w1[4];
    bf16x8 dum[12];                              // (ABL bit 6: the reads land here and nothing waits for them before the MFMAs)
    auto read_a = [&](int buf, int h) {
        if (ABL & 4) return;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + offA[ks] + (buf * 2 + h) * P8_HALF + mt * 4096);
                if (ABL & 64) dum[4 + mt * 4 + ks] = v; else af[mt][ks] = v;
            }
    };
    auto read_w = [&](int buf, int h, bf16x8 (&w)[4]) {
        if (ABL & 4) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + offW[ks] + (buf * 2 + h) * P8_HALF);
            if (ABL & 64) dum[ks] = v; else w[ks] = v;
        }
    };
    auto eat = [&]() {
        if (ABL & 64) {
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("" ::"v"(dum[i]));
        }
    };

    f32x16 acc[4][2];
    // bf16: the bias enters as ONE EXTRA k-step at the start of a tile: W' = [b_hi b_mid b_lo 0 ...] (the fp32 bias as three
    // bf16 terms: their fp32 sum is exact) against A' = [1 1 1 0 ...] -- 8 MFMAs per tile instead of 128 bias registers or
    // adds per lane, and it replaces zeroing the accumulators.  The block's columns never change (the launcher makes the tile
    // order keep n0 per block), so the two operand register sets live for the whole kernel.
    const int n0 = (tile_of_round(0) % nbn) * P8_BN;
    bf16x8 wb[2], ones;
    f32x4 biasr[2];                               // fp32: the lane's bias values in the layout AFTER the transposition
    if (!WORDS) {
        // (H16: the same three-term bias on fp16 operands -- the sum of three fp16 terms is exact for |b| in fp16's normal range)
        auto pack3 = [&](float x0, float x1, float x2) {
            if constexpr (H16) {
                const _Float16 z = (_Float16)0.f;
                return __builtin_bit_cast(bf16x8, f16x8{(_Float16)x0, (_Float16)x1, (_Float16)x2, z, z, z, z, z});
            } else {
                const __bf16 z = (__bf16)0.f;
                return bf16x8{(__bf16)x0, (__bf16)x1, (__bf16)x2, z, z, z, z, z};
            }
        };
        auto rnd = [&](float x) { if constexpr (H16) return (float)(_Float16)x; else return (float)(__bf16)x; };
        const float o = hi ? 0.f : 1.f;
        ones = pack3(o, o, o);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const float b = (p.bias && !hi) ? p.bias[n0 + (wc * 2 + tn) * 32 + li] : 0.f;
            const float b0 = rnd(b), b1 = rnd(b - b0), b2 = rnd(b - b0 - b1);
            wb[tn] = pack3(b0, b1, b2);
            asm volatile("" : "+v"(wb[tn]));     // landed before the pipeline starts: no compiler-inserted wait inside it
        }
    } else {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            biasr[tn] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + (wc * 2 + tn) * 32 + 4 * (lane & 7)) : f32x4{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(biasr[tn]));
        }
    }
    // quadrant (ah, bh): 2 m-tiles x 1 n-tile x 4 k-steps; INIT: first k-tile of an output tile
    auto mma = [&](int ah, int bh, const bf16x8 (&w)[4], auto initc) {
        constexpr bool INIT = decltype(initc)::value;
        if (ABL & 2) return;
        if (!(ABL & 128)) __builtin_amdgcn_s_setprio(1);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (F32) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 wf = __builtin_bit_cast(f32x4, w[ks]);
#pragma unroll
                for (int sft = 0; sft < 4; ++sft)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        float a = __builtin_bit_cast(f32x4, af[mt][ks])[sft];
                        if (RELU) a = fmaxf(a, 0.f);
                        const f32x16 c0 = (INIT && ADD == 0 && ks == 0 && sft == 0) ? zero : acc[2 * ah + mt][bh];
                        acc[2 * ah + mt][bh] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[sft], a, c0, 0, 0, 0);
                    }
            }
        } else if (X3) {
            // w = {hi ks0, hi ks1, lo ks0, lo ks1}; term order per accumulator: w_hi.a_lo, w_lo.a_hi, w_hi.a_hi (small terms
            // first, as gemm_core.h PipeSplitDma); the two accumulators take turns so that no MFMA waits for its predecessor
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ahi[2], alo[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    u32x4 x0 = __builtin_bit_cast(u32x4, af[mt][2 * ks]), x1 = __builtin_bit_cast(u32x4, af[mt][2 * ks + 1]);
                    if (RELU) {          // x < 0  <=>  its hi part is negative: sign bit of the word; the whole word becomes +0
#pragma unroll
                        for (int c = 0; c < 4; ++c) { x0[c] = (unsigned)max((int)x0[c], 0); x1[c] = (unsigned)max((int)x1[c], 0); }
                    }
                    u32x4 h, l;
                    h[0] = __builtin_amdgcn_perm(x0[1], x0[0], 0x07060302u); h[1] = __builtin_amdgcn_perm(x0[3], x0[2], 0x07060302u);
                    h[2] = __builtin_amdgcn_perm(x1[1], x1[0], 0x07060302u); h[3] = __builtin_amdgcn_perm(x1[3], x1[2], 0x07060302u);
                    l[0] = __builtin_amdgcn_perm(x0[1], x0[0], 0x05040100u); l[1] = __builtin_amdgcn_perm(x0[3], x0[2], 0x05040100u);
                    l[2] = __builtin_amdgcn_perm(x1[1], x1[0], 0x05040100u); l[3] = __builtin_amdgcn_perm(x1[3], x1[2], 0x05040100u);
                    ahi[mt] = __builtin_bit_cast(bf16x8, h);
                    alo[mt] = __builtin_bit_cast(bf16x8, l);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[2 * ah + mt][bh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], alo[mt], (INIT && ADD == 0 && ks == 0) ? zero : acc[2 * ah + mt][bh], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[2 * ah + mt][bh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[2 + ks], ahi[mt], acc[2 * ah + mt][bh], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[2 * ah + mt][bh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], ahi[mt], acc[2 * ah + mt][bh], 0, 0, 0);
            }
        } else {
            if (INIT) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[2 * ah + mt][bh] = mfma_h<H16>(wb[bh], ones, ADD != 0 ? acc[2 * ah + mt][bh] : zero);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    bf16x8 a = af[mt][ks];
                    if (RELU) a = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), s16x8{0, 0, 0, 0, 0, 0, 0, 0}));
                    acc[2 * ah + mt][bh] = mfma_h<H16>(w[ks], a, acc[2 * ah + mt][bh]);
                }
        }
        if (!(ABL & 128)) __builtin_amdgcn_s_setprio(0);
        eat();
    };

    // ---- epilogue of one 32-row strip (tm) of the wave tile ----
    // Lane (li, hi) holds row li and columns tn*32 + 8g + 4hi .. +3 (gemm_core.h).  The strip is transposed through a
    // wave-private LDS buffer of 32 rows x 128 bytes (16-byte chunks XOR-swizzled with row & 7) so that lane l then holds
    // 16 consecutive bytes of row 8j + (l >> 3), chunk l & 7: one store instruction writes 8 full 128-byte rows.
    char* const tbuf = smem + P8_TBASE + wave * 4096;
    // (half rows: a lane writes 8 bytes; rows li and li + 8 of a 16-lane write group name the same chunk, so the upper eight
    //  rows take the chunk's OTHER half -- 32 distinct banks per group instead of a 2-way conflict (PMC, round 3: 10-12 % of the
    //  LDS cycles of the half-row launches) -- and the reader swaps the halves back for those rows, a register renaming)
    const int t_wr = li * 128 + (HALF ? (hi ^ ((li >> 3) & 1)) * 8 : 0), t_x = (li & 7) << 4;
    const int t_rd = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const unsigned ldc4 = (unsigned)p.ldc * 4u;
    const unsigned vst = (unsigned)(lane >> 3) * ldc4 + (unsigned)(lane & 7) * 16u;
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((size_t)(p.M - 1) * p.ldc + p.N) * 4), 0x00020000);
    const float cs = p.c_scale;
    const float act_lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    // ACTV (wave-uniform, picked once per strip): 0 = no output activation and c_scale 1, 1 = ReLU and c_scale 1, 2 = the
    // general form.  The first two are what the forward launches; they drop the max / multiply per element (the compiler
    // needs two v_max per fmaxf on values it cannot prove canonical), and the half-row ReLU runs on packed bf16 pairs.
    const int actv = __builtin_amdgcn_readfirstlane(cs != 1.f ? 2 : p.act == ACT_RELU ? 1 : 0);
    auto relu1 = [](float x) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x)); return y; };
    auto epi_t = [&](auto tmc, auto actc, int m0) __attribute__((always_inline)) {
        constexpr int TM = decltype(tmc)::value;
        constexpr int ACTV = decltype(actc)::value;
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const unsigned srow = (unsigned)(m0 + wr * 128 + TM * 32) * ldc4 + (unsigned)(n0 + wc * 64) * (HALF ? 2u : 4u);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = acc[TM][tn][4 * g + c];
                if (!WORDS) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (ACTV == 2) v[c] = fmaxf(v[c], act_lo) * cs;      // (branch-free: ReLU or max with -inf)
                        else if (ACTV == 1 && !HALF) v[c] = relu1(v[c]);
                    }
                }
                if (HALF) {
                    s16x2 w0, w1;
                    if (CF == 3) {
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = __builtin_amdgcn_fmed3f(v[c], -65504.f, 65504.f);
                        w0 = __builtin_bit_cast(s16x2, __builtin_convertvector((f32x2{v[0], v[1]}), f16x2));
                        w1 = __builtin_bit_cast(s16x2, __builtin_convertvector((f32x2{v[2], v[3]}), f16x2));
                    } else {
                        w0 = __builtin_bit_cast(s16x2, __builtin_convertvector((f32x2{v[0], v[1]}), bf16x2));
                        w1 = __builtin_bit_cast(s16x2, __builtin_convertvector((f32x2{v[2], v[3]}), bf16x2));
                    }
                    if (ACTV == 1) {          // ReLU after the rounding: the same values (rounding keeps the sign; -0 becomes +0)
                        w0 = __builtin_elementwise_max(w0, s16x2{0, 0});
                        w1 = __builtin_elementwise_max(w1, s16x2{0, 0});
                    }
                    u32x2 w;
                    w.x = __builtin_bit_cast(unsigned, w0);
                    w.y = __builtin_bit_cast(unsigned, w1);
                    *reinterpret_cast<u32x2*>(tbuf + t_wr + (((tn * 4 + g) << 4) ^ t_x)) = w;
                } else {
                    *reinterpret_cast<f32x4*>(tbuf + t_wr + (((2 * g + hi) << 4) ^ t_x)) = v;
                }
            }
            if (!HALF || tn == 1) {          // the buffer holds 32 rows x 128 bytes: both n-tiles (bf16) or one (fp32)
                u32x4 rj[4];                   // (all four reads in flight before the first store waits for its data)
#pragma unroll
                for (int j = 0; j < 4; ++j) rj[j] = *reinterpret_cast<const u32x4*>(tbuf + t_rd + j * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 r = rj[j];
                    if (HALF && (j & 1)) r = u32x4{rj[j][2], rj[j][3], rj[j][0], rj[j][1]};      // rows 8 j + (lane >> 3): halves stored swapped
                    if (WORDS) {               // lane: row 8j + (l >> 3), columns tn*32 + 4 (l & 7) .. +3
                        f32x4 x = __builtin_bit_cast(f32x4, r) + biasr[tn];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (ACTV == 2) {
                                x[c] = fmaxf(x[c], act_lo);
                                if (X3) x[c] *= cs;
                            } else if (ACTV == 1) {
                                x[c] = relu1(x[c]);
                            }
                        }
                        r = __builtin_bit_cast(u32x4, x);
                        if (CF == 1) {         // split pairs (common.h pack_split), two elements at a time
#pragma unroll
                            for (int c = 0; c < 4; c += 2) {
                                const f32x2 v2 = {x[c], x[c + 1]};
                                const bf16x2 h2 = __builtin_convertvector(v2, bf16x2);
                                const bf16x2 l2 = __builtin_convertvector(v2 - __builtin_convertvector(h2, f32x2), bf16x2);
                                const unsigned hb = __builtin_bit_cast(unsigned, h2), lb = __builtin_bit_cast(unsigned, l2);
                                r[c] = __builtin_amdgcn_perm(hb, lb, 0x05040100u);
                                r[c + 1] = __builtin_amdgcn_perm(hb, lb, 0x07060302u);
                            }
                        }
                    }
                    if (!(ABL & 8)) {
                        __builtin_amdgcn_raw_buffer_store_b128(r, rc, vst, srow + (unsigned)(8 * j) * ldc4 + (HALF ? 0u : (unsigned)tn * 128u), 0);
                        // A 128-bit buffer store reads its data registers a few cycles after issue; hipcc pads a following VALU
                        // write of them ONLY when the store has no SGPR soffset (LLVM's hazard rule assumes the register form is
                        // safe) -- on gfx950 it is not: split-pair epilogues, whose pack code reuses the registers at once, stored
                        // garbage in one dword of some lanes until this pad went in (found with the bit-identity test).
                        // (the data registers are named as an input so that they stay allocated across the pad: a side-effecting asm
                        //  orders memory operations only, and `r` is otherwise dead after the store)
                        asm volatile("s_nop 3" ::"v"(r) : "memory");
                    } else
                        asm volatile("" ::"v"(r));
                }
            }
        }
    };
    auto epi = [&](auto tmc, int m0) __attribute__((always_inline)) {     // (not inlined = the accumulators live in scratch)
        if (actv == 0) epi_t(tmc, std::integral_constant<int, 0>{}, m0);
        else if (actv == 1) epi_t(tmc, std::integral_constant<int, 1>{}, m0);
        else epi_t(tmc, std::integral_constant<int, 2>{}, m0);
    };

    // ---- prologue: the six half-tiles whose staging phase lies before the first compute phase ----
    ld2(ra, sdst, vA, c0.sa + kti(c0) * KA, 128u * lda4);                                         // A_0(0)
    ldw(sdst + P8_WBASE, c0.sw + kti(c0) * KW, 0);                                                // W_0(0)
    ldw(sdst + P8_WBASE + P8_HALF, c0.sw + kti(c0) * KW, 1);                                      // W_1(0)
    ld2(ra, sdst + P8_HALF, vA, c0.sa + kti(c0) * KA + 64u * lda4, 128u * lda4);                  // A_1(0)
    ld2(ra, sdst + 2 * P8_HALF, vA, c1.sa + kti(c1) * KA, 128u * lda4);                           // A_0(1)
    ldw(sdst + P8_WBASE + 2 * P8_HALF, c1.sw + kti(c1) * KW, 0);                                  // W_0(1)
    p8_wait_vm<8>();
    p8_barrier();
    if (wr == 1) p8_barrier();                       // the two wave rows run one barrier apart from here on

    // One K-tile = four phases on buffer B; c1 / c2 = the K-tiles one and two ahead.  VAR places the epilogue strips
    // (pm0: row base of the tile they belong to) and sets the counted waits: the half-tile a phase's wait must retire was
    // issued four load parts earlier, with 8 LDS-direct loads and the epilogue stores of the parts since behind it.
    auto ktile = [&](auto bufc, auto varc, int pm0) {
        constexpr int B = decltype(bufc)::value;
        constexpr int VAR = decltype(varc)::value;
        // (a load part is [strip: E stores][2 LDS-direct loads][wait]: the wait of part j lets the loads of parts j-3 .. j
        //  and the stores of those four parts stay in flight; strips sit in L3, L4, F1, F2)
        constexpr int N1 = 8 + (VAR == P8_FIRST ? 3 * E : VAR == P8_SECOND ? E : 0);
        constexpr int N2 = 8 + (VAR == P8_FIRST ? 4 * E : 0);
        constexpr int N4 = 8 + (VAR == P8_FIRST || VAR == P8_LAST ? 2 * E : 0);
        using Init = std::integral_constant<bool, VAR == P8_FIRST || VAR == P8_FIRST0>;
        const unsigned a1 = c1.sa + kti(c1) * KA, w1o = c1.sw + kti(c1) * KW;
        const unsigned a2 = c2.sa + kti(c2) * KA, w2o = c2.sw + kti(c2) * KW;
        // phase 1: quadrant (a0, w0)
        if (VAR == P8_FIRST) {               // (before the fragment reads: the strip's temporaries need their registers)
            epi(std::integral_constant<int, 2>{}, pm0);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_w(B, 0, w0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(B, 0);
        stage_w(B ^ 1, 1, w1o);
        p8_wait_vm<N1>();
        if (!(ABL & 32)) p8_barrier();
        if (!(ABL & 64)) p8_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0, w0, Init{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 48)) p8_barrier();
        // phase 2: quadrant (a0, w1)
        if (VAR == P8_FIRST) {
            epi(std::integral_constant<int, 3>{}, pm0);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_w(B, 1, w1);
        stage_a(B ^ 1, 1, a1);
        p8_wait_vm<N2>();
        if (!(ABL & 32)) p8_barrier();
        if (!(ABL & 64)) p8_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1, w1, Init{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 48)) p8_barrier();
        // phase 3: quadrant (a1, w1)
        if (VAR == P8_LAST) {
            epi(std::integral_constant<int, 0>{}, pm0);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_a(B, 1);
        stage_a(B, 0, a2);
        if (!(ABL & 32)) p8_barrier();
        if (!(ABL & 64)) p8_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1, w1, Init{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 48)) p8_barrier();
        // phase 4: quadrant (a1, w0)
        if (VAR == P8_LAST) {
            epi(std::integral_constant<int, 1>{}, pm0);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage_w(B, 0, w2o);
        p8_wait_vm<N4>();
        if (!(ABL & 32)) p8_barrier();
        if (!(ABL & 64)) p8_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0, w0, Init{});
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 48)) p8_barrier();
        c1 = c2;
        advance(c2);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using Mid = std::integral_constant<int, P8_MID>;

    int pm0 = 0;
    for (int round = 0;; ++round) {
        const int v = tile_of_round(round);
        if (v >= n_tiles) break;
        const int m0 = (v / nbn) * P8_BM;
        if (ADD == 6 && (ABL & 256)) {
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = min(m0 + wr * 128 + tm * 32 + g * 8 + (lane >> 3), p.M - 1);
                    const float* r0 = p.g0 + (size_t)p.gi0[row] * p.ldg0 + n0 + wc * 64 + (lane & 7) * 4;
                    const float* r1 = p.g1 + (size_t)p.gi1[row] * p.ldg1 + n0 + wc * 64 + (lane & 7) * 4;
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) {
                        const f32x4 x = *reinterpret_cast<const f32x4*>(r0 + tn * 32) + *reinterpret_cast<const f32x4*>(r1 + tn * 32);
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[tm][tn][4 * g + c] = x[c];
                    }
                }
        } else if (ADD != 0 && !(ABL & 512))
            tile_init<4, 2, ADD, F32>(p, m0, n0, wr, wc, lane, acc);   // (PLAIN only for fp32: formats fold away)
        if (SPREAD && round > 0) {                   // strips 2 / 3 of the previous tile go out under this tile's first phases
            ktile(B0{}, std::integral_constant<int, P8_FIRST>{}, pm0);
            ktile(B1{}, std::integral_constant<int, P8_SECOND>{}, pm0);
        } else {
            ktile(B0{}, std::integral_constant<int, P8_FIRST0>{}, 0);
            ktile(B1{}, Mid{}, 0);
        }
        for (int kt = 2; kt < KT - 2; kt += 2) {     // (KT is even and >= 4: the launcher checks K % 128 == 0, K >= 256)
            ktile(B0{}, Mid{}, 0);
            ktile(B1{}, Mid{}, 0);
        }
        ktile(B0{}, Mid{}, 0);
        ktile(B1{}, std::integral_constant<int, P8_LAST>{}, m0);
        if (!SPREAD) {                               // additive operands are loaded as accumulator inits of the next tile
            epi(std::integral_constant<int, 2>{}, m0);
            epi(std::integral_constant<int, 3>{}, m0);
        }
        pm0 = m0;
    }
    if (wr == 0) p8_barrier();
    if (SPREAD) {
        epi(std::integral_constant<int, 2>{}, pm0);
        epi(std::integral_constant<int, 3>{}, pm0);
    }
    p8_wait_vm<0>();                                 // no LDS-direct load may outlive the block's LDS allocation
}

}  // namespace

// full rounds of a large-M launch on 256 x 256 tiles: half-row bf16 operands (prec 1), exact fp32 (prec 0) or split-bf16
// on split-pair operands (prec 3); 1 = operand combination not built (the caller falls back to the older kernels)
int launch_gemm_p8(const GemmArgs& a, int n_tiles, int grid, hipStream_t s) {
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
    const bool f32 = a.prec == 0, x3 = a.prec == 3;
    if (a.rowscale || a.act == ACT_SIGMOID || (add != 0 && add != 1 && add != 6)) return 1;
    // the fp32 / split-bf16 epilogues read the bias as float4 (the half-row one as scalars): an unaligned bias pointer of a
    // caller of vlsat_k_gemm goes to the older kernels, which have the scalar fallback
    if ((f32 || x3) && a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return 1;
    // (round 4, measured and dropped: loading the accumulator inits of the wave tile's second row half one phase later, under
    //  the MFMAs of phases 1 / 2 -- fp32 nn_edge.0 + gathered rows 923 vs 917 us, out-projection + residual 480 vs 483: the
    //  cost of additive operands is not latency at the start of a tile; it is consistent with every CU pulling its 256-512 KB at the same
    //  moment: 64-128 MB per round of tiles at what the memory system delivers)
    if (f32 ? (a.a_split || a.c_split || a.r_split || a.c_scale != 1.f)
            : x3 ? (a.a_split != 1 || a.c_split == 2 || !a.Wlo) : (a.prec != 1 || a.a_split != 2 || a.c_split == 1)) return 1;
    const int kt = (f32 || x3) ? 32 : P8_BK;          // an output tile is an even number (>= 4) of K-tiles
    // (M need not be a multiple of the tile: the last panel's rows past M are outside every buffer descriptor -- loads return zeros,
    //  stores are dropped -- and tile_init clamps the row of an additive operand)
    if (a.N % P8_BN || a.K % (2 * kt) || a.K < 4 * kt || n_tiles > (long)((a.M + P8_BM - 1) / P8_BM) * (a.N / P8_BN)) return 1;
    const int nbn = a.N / P8_BN;
    if (grid % 8 || (grid / 8) % nbn) return 1;      // the kernel keeps one column tile per block (bias registers)
#define VLSAT_P8(MODE, ADD, RELU, CF) hipLaunchKernelGGL((gemm_p8_kernel<MODE, ADD, RELU, CF>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn)
#define VLSAT_P8_ABL(X) hipLaunchKernelGGL((gemm_p8_kernel<0, 0, false, 2, X>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn)
    const bool c16 = a.c_f16_cols > 0;            // (the whole output as fp16 half rows: half-row launches only)
    if (c16 && (a.c_f16_cols != a.N || f32 || x3 || a.c_split)) return 1;
    if (a.half_f16 && (f32 || x3 || a.c_split)) return 1;         // (fp16 operands: half-row launches; their half-row outputs come as c_f16_cols == N)
    const int key = add * 4 + (a.relu_a ? 2 : 0) + (a.c_split ? 1 : 0);
    if (f32) {
        switch (key) {
            case 0: VLSAT_P8(1, 0, false, 0); break;
            case 2: VLSAT_P8(1, 0, true, 0); break;
            case 4: VLSAT_P8(1, 1, false, 0); break;
            case 24: VLSAT_P8(1, 6, false, 0); break;
            case 26: VLSAT_P8(1, 6, true, 0); break;
            default: return 1;
        }
    } else if (x3) {
        switch (key) {
            case 0: VLSAT_P8(2, 0, false, 0); break;
            case 1: VLSAT_P8(2, 0, false, 1); break;
            case 2: VLSAT_P8(2, 0, true, 0); break;
            case 3: VLSAT_P8(2, 0, true, 1); break;
            case 4: VLSAT_P8(2, 1, false, 0); break;
            case 25: VLSAT_P8(2, 6, false, 1); break;
            case 27: VLSAT_P8(2, 6, true, 1); break;
            default: return 1;
        }
#ifdef VLSAT_EXPERIMENTS
    } else if (a.ablate && key == 27) {               // timing experiments on the gathered-row launch
        switch (a.ablate) {
            case 1: hipLaunchKernelGGL((gemm_p8_kernel<0, 6, true, 2, 256>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            default: hipLaunchKernelGGL((gemm_p8_kernel<0, 6, true, 2, 512>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
        }
    } else if (a.ablate && key == 1) {                // timing experiments (tools/p8_check.py --ablate)
        switch (a.ablate) {
            case 1: VLSAT_P8_ABL(1); break;
            case 2: VLSAT_P8_ABL(2); break;
            case 3: VLSAT_P8_ABL(3); break;
            case 4: VLSAT_P8_ABL(4); break;
            case 6: VLSAT_P8_ABL(6); break;
            case 7: VLSAT_P8_ABL(7); break;
            case 8: VLSAT_P8_ABL(8); break;
            case 5: VLSAT_P8_ABL(5); break;
            case 37: VLSAT_P8_ABL(37); break;
            case 65: VLSAT_P8_ABL(65); break;
            default: VLSAT_P8_ABL(15); break;
        }
#endif
    } else {
        if (a.half_f16) {                     // MODE 3: fp16 operands; output fp32 (CF 0) or fp16 half rows (CF 3)
            switch (add * 4 + (a.relu_a ? 2 : 0) + (c16 ? 1 : 0)) {
                case 0: VLSAT_P8(3, 0, false, 0); break;
                case 1: VLSAT_P8(3, 0, false, 3); break;
                case 2: VLSAT_P8(3, 0, true, 0); break;
                case 3: VLSAT_P8(3, 0, true, 3); break;
                case 25: VLSAT_P8(3, 6, false, 3); break;
                case 27: VLSAT_P8(3, 6, true, 3); break;
                default: return 1;
            }
        } else
        switch (key) {
            case 0: if (c16) VLSAT_P8(0, 0, false, 3); else VLSAT_P8(0, 0, false, 0); break;
            case 1: VLSAT_P8(0, 0, false, 2); break;
            case 2: VLSAT_P8(0, 0, true, 0); break;
            case 3: VLSAT_P8(0, 0, true, 2); break;
            case 4: VLSAT_P8(0, 1, false, 0); break;
            case 5: VLSAT_P8(0, 1, false, 2); break;
            case 25: VLSAT_P8(0, 6, false, 2); break;
            case 27: VLSAT_P8(0, 6, true, 2); break;
            default: return 1;
        }
    }
#undef VLSAT_P8_ABL
#undef VLSAT_P8
    if (a.launches) ++*a.launches;
    VLSAT_LAUNCH_CHECK("gemm_p8");
    return 0;
}

}  // namespace vlsat
