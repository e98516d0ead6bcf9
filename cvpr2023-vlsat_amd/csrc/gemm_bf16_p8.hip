// bf16 GEMM of the single-rounding modes for the large edge-row launches (BASELINE configs[2]) -- same contract as
// gemm_f32.hip / gemm_bf16_ring.hip,
//     C[M,N] = act((A[M,K] . W[N,K]^T) + bias + resid_scale*resid + g0[gi0] + g1[gi1]) * c_scale,
// for A stored as HALF ROWS (bf16 at byte 2 k of an fp32-pitched row) and one bf16 weight plane.  Every nn.Linear on edge
// rows goes through it in the bf16_mixed / bf16 modes (reference network_MMG.py:59-79,93-98; attention.py:54-58,77;
// network_PointNet.py:328-341).
//
// Structure (cdna_hip_programming.md "The 256^2 8-phase template", rebuilt for this library's operand formats):
//   * ONE persistent block of 8 waves per CU, block tile 256 x 256, waves 2 (M) x 4 (N), wave tile 128 x 64 =
//     4 x 2 tiles of v_mfma_f32_32x32x16_bf16 (transposed product: a lane owns one output ROW, tile_epilogue's layout).
//   * K in tiles of 64; LDS holds two K-tiles (2 x 64 KB), each as four HALF-TILES of 128 rows x 128 bytes:
//       A_h = rows {wr*128 + h*64 + i} of the block's A panel,  W_h = rows {wc*64 + h*32 + j} of its weight panel,
//     i.e. half-tile h is what quadrant h of EVERY wave reads.  Rows are 128 bytes, 16-byte chunks XOR-swizzled with
//     (row >> 1) & 7 on the global side of the LDS-direct loads and on the fragment reads (0 bank conflicts).
//   * A K-tile is FOUR PHASES, one per quadrant of the wave tile, in the order (a0,w0) (a0,w1) (a1,w1) (a1,w0); a phase is
//         ds_read the operands the quadrant does not hold yet (12 / 4 / 8 / 0 ds_read_b128)
//         issue ONE half-tile of LDS-direct loads (2 x buffer_load_dwordx4 ... lds per lane)
//         s_waitcnt vmcnt(8) ; s_barrier ; s_waitcnt lgkmcnt(0) ; 8 MFMAs under s_setprio 1 ; s_barrier
//     and the two wave rows run one barrier apart, so on every SIMD one wave is in its MFMA cluster while its partner
//     reads LDS and issues loads.
//   * Loads are never drained: phase q of K-tile g stages  W_1(g+1), A_1(g+1), A_0(g+2), W_0(g+2)  for q = 1..4 -- each
//     into the buffer half its previous tenant's last ds_read left >= 2 phases earlier -- and every wait is the counted
//     vmcnt(8): "everything but the four half-tiles issued last has landed", which is exactly what the next phase reads.
//     The sequence of K-tiles runs on across output tiles (the staging cursor is two K-tiles ahead, in the next tile if
//     need be).
// Results are bit-identical to the 128 x 128 kernel's (same k order per accumulator).
// Roofline: bf16 MFMA (2.5 PF dense); per K-tile a block moves 64 KB from L2 for 8.4 MFLOP.
#include <type_traits>

#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

template <int N> __device__ __forceinline__ void p8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void p8_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void p8_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 64;
constexpr int P8_HALF = 128 * 128;            // bytes of a half-tile
constexpr int P8_WBASE = 4 * P8_HALF;         // weight half-tiles start after the four A half-tiles (2 buffers x 2 halves)

// ABL (timing experiments, results are garbage): bit 0 no LDS-direct loads after the prologue, bit 1 no MFMAs, bit 2 no
// fragment reads, bit 3 no epilogue
template <int ADD, bool RELU, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_p8_kernel(GemmArgs p, int n_tiles, int nbn) {
    __shared__ __attribute__((aligned(16))) char smem[8 * P8_HALF];
    typedef short s16x8 __attribute__((ext_vector_type(8)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int li = lane & 31, hi = lane >> 5;
    const int g8 = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int KT = p.K / P8_BK;
    auto tile_of_round = [&](int r) { return (r * 8 + xcd) * g8 + slot; };
    if (tile_of_round(0) >= n_tiles) return;

    // ---- LDS-direct loader: lane constants (a wave instruction fills 8 rows x 128 bytes) ----
    const int srow = wave * 8 + (lane >> 3);                                   // LDS row inside a 64-row round
    const unsigned schunk = (unsigned)((lane & 7) ^ ((srow >> 1) & 7));          // logical chunk this lane fetches
    const unsigned lda4 = (unsigned)p.lda * 4u, ldw2 = (unsigned)p.ldw * 2u;
    const unsigned vA = (unsigned)srow * lda4 + schunk * 16u;
    const unsigned vW = (unsigned)((wave >> 2) * 64 + (wave & 3) * 8 + (lane >> 3)) * ldw2 + schunk * 16u;
    const int na = (int)((size_t)(p.M - 1) * lda4 + (size_t)p.K * 2), nw = (int)(((size_t)(p.N - 1) * p.ldw + p.K) * 2);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, na, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.Whi), 0, nw, 0x00020000);
    char* const sdst = smem + wave * 1024;
    // half-tile h of the K-tile at byte offsets (sa, sw) into buffer `buf`
    auto stage_a = [&](int buf, int h, unsigned sa) {
        char* d = sdst + (buf * 2 + h) * P8_HALF;
        const unsigned s0 = sa + (unsigned)(h * 64) * lda4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, d, 16, vA, s0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, d + 8192, 16, vA, s0 + 128u * lda4, 0, 0);
    };
    auto stage_w = [&](int buf, int h, unsigned sw) {
        char* d = sdst + P8_WBASE + (buf * 2 + h) * P8_HALF;
        const unsigned s0 = sw + (unsigned)(h * 32) * ldw2;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, d, 16, vW, s0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, d + 8192, 16, vW, s0 + 128u * ldw2, 0, 0);
    };

    // ---- staging cursor: the K-tile sequence of this block, across output tiles ----
    // (past the block's last tile the cursor stays on it: the loads keep their rhythm, so every wait is the same counted
    //  one and the K loop has no branch; what they fetch is never read)
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
    Cur c1{0, 0, 0u, 0u};
    locate(c1);
    Cur c0 = c1;
    advance(c1);
    Cur c2 = c1;
    advance(c2);

    // ---- fragment readers: per-lane byte offsets of k-step ks (the chunk XOR is not an add: one register per k-step) ----
    const int swz = (li >> 1) & 7;
    int offA[4], offW[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int chunk = ((2 * ks + hi) ^ swz) * 16;
        offA[ks] = (wr * 64 + li) * 128 + chunk;
        offW[ks] = P8_WBASE + (wc * 32 + li) * 128 + chunk;
    }
    bf16x8 af[2][4], w0[4], w1[4];
    auto read_a = [&](int buf, int h) {
        if (ABL & 4) return;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s16x8 v = *reinterpret_cast<const s16x8*>(smem + offA[ks] + (buf * 2 + h) * P8_HALF + mt * 4096);
                af[mt][ks] = __builtin_bit_cast(bf16x8, v);
            }
    };
    auto read_w = [&](int buf, int h, bf16x8 (&w)[4]) {
        if (ABL & 4) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(smem + offW[ks] + (buf * 2 + h) * P8_HALF);
    };

    f32x16 acc[4][2];
    zero_acc<4, 2>(acc);
    // quadrant (ah, bh): 2 m-tiles x 1 n-tile x 4 k-steps
    auto mma = [&](int ah, int bh, const bf16x8 (&w)[4]) {
        if (ABL & 2) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x8 a = af[mt][ks];
                if (RELU) a = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, a), s16x8{0, 0, 0, 0, 0, 0, 0, 0}));
                acc[2 * ah + mt][bh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], a, acc[2 * ah + mt][bh], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: the six half-tiles whose staging phase lies before the first compute phase ----
    stage_a(0, 0, c0.sa);
    stage_w(0, 0, c0.sw);
    stage_w(0, 1, c0.sw);
    stage_a(0, 1, c0.sa);
    stage_a(1, 0, c1.sa + c1.kt * 128u);
    stage_w(1, 0, c1.sw + c1.kt * 128u);
    p8_wait_vm<8>();
    p8_barrier();
    if (wr == 1) p8_barrier();                       // the two wave rows run one barrier apart from here on

    // One K-tile = four phases on buffer B; c1 / c2 = the K-tiles one and two ahead.
    auto ktile = [&](auto bufc) {
        constexpr int B = decltype(bufc)::value;
        constexpr bool ST = !(ABL & 1);
        const unsigned a1 = c1.sa + c1.kt * 128u, w1o = c1.sw + c1.kt * 128u;
        const unsigned a2 = c2.sa + c2.kt * 128u, w2o = c2.sw + c2.kt * 128u;
        // phase 1: quadrant (a0, w0)
        read_w(B, 0, w0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(B, 0);
        if (ST) stage_w(B ^ 1, 1, w1o);
        p8_wait_vm<8>();
        p8_barrier();
        p8_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0, w0);
        __builtin_amdgcn_sched_barrier(0);
        p8_barrier();
        // phase 2: quadrant (a0, w1)
        read_w(B, 1, w1);
        if (ST) stage_a(B ^ 1, 1, a1);
        p8_wait_vm<8>();
        p8_barrier();
        p8_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 1, w1);
        __builtin_amdgcn_sched_barrier(0);
        p8_barrier();
        // phase 3: quadrant (a1, w1)
        read_a(B, 1);
        if (ST) stage_a(B, 0, a2);
        p8_barrier();
        p8_wait_lds();
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1, w1);
        __builtin_amdgcn_sched_barrier(0);
        p8_barrier();
        // phase 4: quadrant (a1, w0)
        if (ST) stage_w(B, 0, w2o);
        p8_wait_vm<8>();
        p8_barrier();
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0, w0);
        __builtin_amdgcn_sched_barrier(0);
        p8_barrier();
        c1 = c2;
        advance(c2);
    };

    // ---- epilogue of one quadrant: bias, activation, scale, store (tiles of this kernel are always interior) ----
    // Lane (li, hi) holds row li of a 32 x 32 tile and columns 8 g + 4 hi .. + 3, g = 0..3 (gemm_core.h).  Stores go
    // through a buffer descriptor: one per-lane offset register for the whole kernel, the rest is scalar + immediate.
    const int half_out = p.c_split == 2;
    const unsigned crow = (unsigned)(wr * 128 + li) * (unsigned)p.ldc * 4u + (unsigned)hi * (half_out ? 8u : 16u);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((size_t)(p.M - 1) * p.ldc + p.N) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.C), 0, p.N * 4, 0x00020000);
    auto epilogue = [&](int m0, int n0, int ah, int bh) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const int ncol = n0 + (wc * 2 + bh) * 32;                       // first column of the quadrant's n-tile
        f32x4 b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)hi * 16u, (unsigned)(ncol + 8 * g) * 4u, 0));
        }
        const float cs = p.c_scale;
        const int act = p.act;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const unsigned srow = (unsigned)(m0 + (2 * ah + mt) * 32) * (unsigned)p.ldc * 4u;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = acc[2 * ah + mt][bh][4 * g + c] + b[g][c];
                if (act == ACT_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                } else if (act == ACT_SIGMOID) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = 1.f / (1.f + __expf(-v[c]));
                }
                v *= cs;
                if (half_out) {
                    u32x2 w;
                    w.x = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16);
                    w.y = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16);
                    __builtin_amdgcn_raw_buffer_store_b64(w, rc, crow, srow + (unsigned)(ncol + 8 * g) * 2u, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rc, crow,
                                                           srow + (unsigned)(ncol + 8 * g) * 4u, 0);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[2 * ah + mt][bh][4 * g + c] = 0.f;
            }
        }
    };

    for (int round = 0;; ++round) {
        const int v = tile_of_round(round);
        if (v >= n_tiles) break;
        const int m0 = (v / nbn) * P8_BM, n0 = (v % nbn) * P8_BN;
        if (ADD != 0) tile_init<4, 2, ADD>(p, m0, n0, wr, wc, lane, acc);
        for (int kt = 0; kt < KT; kt += 2) {         // (KT is even: the launcher checks K % 128 == 0)
            ktile(std::integral_constant<int, 0>{});
            ktile(std::integral_constant<int, 1>{});
        }
        if (!(ABL & 8)) {
            epilogue(m0, n0, 0, 0);
            epilogue(m0, n0, 0, 1);
            epilogue(m0, n0, 1, 1);
            epilogue(m0, n0, 1, 0);
        }
    }
    if (wr == 0) p8_barrier();
    p8_wait_vm<0>();                                 // no LDS-direct load may outlive the block's LDS allocation
}

}  // namespace

// full rounds of a large-M half-row launch on 256 x 256 tiles; 1 = operand combination not built (caller falls back)
int launch_gemm_p8(const GemmArgs& a, int n_tiles, int grid, hipStream_t s) {
    const int add = (a.resid ? 1 : 0) | (a.g0 ? 2 : 0) | (a.g1 ? 4 : 0);
    if (a.prec != 1 || a.a_split != 2 || a.c_split == 1 || a.rowscale || (add != 0 && add != 1 && add != 6)) return 1;
    if (a.N % P8_BN || a.K % 128 || a.M % P8_BM) return 1;
    const int nbn = a.N / P8_BN;
#define VLSAT_P8(ADD, RELU) hipLaunchKernelGGL((gemm_p8_kernel<ADD, RELU>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn)
#define VLSAT_P8_ADD(RELU)                  \
    switch (add) {                          \
        case 0: VLSAT_P8(0, RELU); break;   \
        case 1: VLSAT_P8(1, RELU); break;   \
        default: VLSAT_P8(6, RELU); break;  \
    }
    if (a.ablate && add == 0 && !a.relu_a) {          // timing experiments (tools/p8_check.py --ablate)
        switch (a.ablate) {
            case 1: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 1>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            case 2: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 2>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            case 3: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 3>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            case 4: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 4>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            case 6: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 6>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            case 7: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 7>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            case 8: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 8>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
            default: hipLaunchKernelGGL((gemm_p8_kernel<0, false, 15>), dim3(grid), dim3(512), 0, s, a, n_tiles, nbn); break;
        }
    } else if (a.relu_a) { VLSAT_P8_ADD(true) } else { VLSAT_P8_ADD(false) }
#undef VLSAT_P8_ADD
#undef VLSAT_P8
    if (a.launches) ++*a.launches;
    VLSAT_LAUNCH_CHECK("gemm_bf16_p8");
    return 0;
}

}  // namespace vlsat
