// NOT part of the library (source only, kept for the record of round 5): the "ping-pong" variant of the eight-wave bf16 edge-attention
// kernel of csrc/flash_attn_bf16.hip.  It was built into the library behind vlsat_debug_option "flash_pp", produced outputs BIT-IDENTICAL
// to the shipped kernel (tools/flash_pp_check.py at the commit that adds this file) and ran 4 % SLOWER at the cfg 5 scene
// (14.17-14.27 vs 13.57-13.82 ms per scene, three alternations on one box): enforcing the anti-phase of two wave groups with a barrier per
// segment does not make the softmax of one wave run under the MFMAs of another.  What bounds the kernel instead:
// profiles/r05_probes/flash_ablate_matrix.txt.  To build it again: paste the kernel into the anonymous namespace of flash_attn_bf16.hip
// (it uses that file's helpers: max3f, pk_add, pk_sub, fb_wait_vm, FB_KV, FB_VSUB) and launch it with 512 threads per 256-query tile.
typedef s16x4 fp_s16x4;

// ---- "Ping-pong" variant of the eight-wave kernel (round 5; half rows, single rounding, head dim 64, no split keys) ----
// Counters of the kernel above (profiles/r05_probes/flash_pipe.md): per wave and 64-key tile 512 matrix-pipe cycles and 672 VALU
// cycles, and the SIMD time per wave-tile is their SUM -- the waves of a CU fall into step (they meet at a barrier per tile and
// queue for the same pipe in the same order), so the softmax of one wave never runs under the MFMAs of another.  Here the anti-phase
// is enforced instead of hoped for (cdna_hip_programming.md, the 8-wave attention structure): a wave's work is cut into
//     M segment u:  O += V(u-1)^T . P(u-1)  and  S(u) = K(u) . Q^T      (16 MFMAs, the LDS fragment reads, the tile loads)
//     V segment u:  online softmax of S(u) -> P(u), fp32 in the score registers (VALU only; the M segment rounds P to bf16 chunk by chunk)
// with ONE block-wide barrier after every segment, and waves 4..7 (group B) run one segment behind waves 0..3 (group A): B enters
// through an extra barrier and A leaves through one.  Every SIMD holds one A and one B wave of the block, so at any time one of
// them is in its MFMA segment and the other in its VALU segment.
// K / V tiles: ring of FOUR buffers (it fits under the 68 KB the output transposition needs anyway).  Tile w is read by A in
// segments 2w (K) and 2w + 2 (V) and by B one segment later, so its buffer is free for tile w + 4 from segment 2w + 4 on: a wave
// issues its share of tile u + 2 at the start of its M segment u (A: segment 2u, B: 2u + 1), and every segment ends with "all
// but the newest tile's loads have landed" + barrier, which is what the next segment of EITHER group reads.  All LDS reads are
// inline asm with counted lgkmcnt (hipcc drains vmcnt in front of an LDS read it cannot tell from the LDS-direct loads in flight),
// buffer slots are compile-time constants (loop unrolled by four).
constexpr int PP_NB = 4;
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void flash_attn_pp_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, int ldq, int ldkv, int ldo, const int4* __restrict__ tiles, int n_tiles) {
    constexpr int D = 64, ROWB = 2 * D, KBYTES = FB_KV * ROWB, VPLANE = 4 * FB_VSUB, BUF = KBYTES + VPLANE, OPITCH = D + 4;
    constexpr int SMEM = PP_NB * BUF > 8 * 32 * OPITCH * 4 ? PP_NB * BUF : 8 * 32 * OPITCH * 4;
    __shared__ __attribute__((aligned(16))) char smem[SMEM];

    const int tile_id = xcd_remap(blockIdx.x, n_tiles);
    const int4 t = tiles[tile_id];
    const int row_base = t.x, n_tok = t.y, q0 = t.z, head = t.w;
    const int n = (n_tok + FB_KV - 1) / FB_KV;                      // key tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                      // 0 = A, 1 = B (one segment behind)
    const int li = lane & 31, hi = lane >> 5;
    const size_t col0 = (size_t)head * D;
    const bool wave_active = q0 + wave * 32 < n_tok;
    int qrow = q0 + wave * 32 + li;
    if (qrow >= n_tok) qrow = n_tok - 1;

    bf16x8 qh[4];
    {
        const char* qrowp = reinterpret_cast<const char*>(Q + (size_t)(row_base + qrow) * ldq) + col0 * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qh[ks] = *reinterpret_cast<const bf16x8*>(qrowp + (8 * hi + 16 * ks) * 2);
    }
    f32x16 o[2], s[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; s[0][r] = 0.f; s[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    // LDS-direct loads: a wave's share of a tile is ONE K instruction (8 rows x 128 B) and ONE V instruction (32 keys x 32 B of one sub-tile)
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K + (size_t)row_base * ldkv), 0, (int)(unsigned)((size_t)n_tok * ldkv * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V + (size_t)row_base * ldkv), 0, (int)(unsigned)((size_t)n_tok * ldkv * 4), 0x00020000);
    const unsigned ld4 = (unsigned)ldkv * 4u;
    const int krow = wave * 8 + (lane >> 3);
    const unsigned vK = (unsigned)krow * ld4 + (unsigned)col0 * 2u + (unsigned)(((lane & 7) ^ ((krow >> 1) & 7)) << 4);
    const unsigned vV = (unsigned)(lane >> 1) * ld4 + (unsigned)col0 * 2u + (unsigned)(lane & 1) * 16u;
    const int vsub = wave >> 1, vkh = wave & 1;
    auto dma_tile = [&](int kv0, char* buf) {
        const unsigned s0 = (unsigned)kv0 * ld4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, buf + wave * 1024, 16, vK, s0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, buf + KBYTES + vsub * FB_VSUB + vkh * 1024, 16, vV, s0 + (unsigned)vkh * 32u * ld4 + (unsigned)vsub * 32u, 0, 0);
    };
    // lane constants of the fragment reads (everything else is an immediate)
    const unsigned lds0 = (unsigned)(uintptr_t)((char __attribute__((address_space(3)))*)smem);
    const unsigned offK = lds0 + (unsigned)(li * ROWB);               // row li of a K image; rows li, li + 32 share the swizzle
    const unsigned kswz = (unsigned)((li >> 1) & 7);
    unsigned offKs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) offKs[ks] = offK + ((((unsigned)hi + 2u * ks) ^ kswz) << 4);
    const unsigned offV = lds0 + (unsigned)(KBYTES + ((lane >> 4) & 1) * FB_VSUB + (4 * hi + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);

    dma_tile(0, smem);
    if (n > 1) { dma_tile(FB_KV, smem + BUF); fb_wait_vm<2>(); } else fb_wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    if (grp) asm volatile("s_barrier" ::: "memory");

#define PP_READ_V(J, S)                                                                                                        \
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"                               \
                 "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"                                    \
                 : "=&v"(vx##S##0), "=&v"(vx##S##1), "=&v"(vx##S##2), "=&v"(vx##S##3)                                          \
                 : "v"(offV), "n"(SV * BUF + (J) * 512), "n"(SV * BUF + (J) * 512 + 256), "n"(SV * BUF + (J) * 512 + 2 * FB_VSUB), "n"(SV * BUF + (J) * 512 + 2 * FB_VSUB + 256))
#define PP_USE_V(S, N)                                                                                                         \
    do {                                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(vx##S##0), "+v"(vx##S##1), "+v"(vx##S##2), "+v"(vx##S##3));              \
        vf0 = __builtin_shufflevector(__builtin_bit_cast(bf16x4, vx##S##0), __builtin_bit_cast(bf16x4, vx##S##1), 0, 1, 2, 3, 4, 5, 6, 7); \
        vf1 = __builtin_shufflevector(__builtin_bit_cast(bf16x4, vx##S##2), __builtin_bit_cast(bf16x4, vx##S##3), 0, 1, 2, 3, 4, 5, 6, 7); \
    } while (0)
#define PP_PV(J)                                                                                                     \
    do {                                                                                                             \
        f32x4 p0, p1;                                                                                                \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) { p0[c] = s[(J) >> 1][8 * ((J) & 1) + c]; p1[c] = s[(J) >> 1][8 * ((J) & 1) + 4 + c]; } \
        const bf16x8 ph = __builtin_shufflevector(__builtin_convertvector(p0, bf16x4), __builtin_convertvector(p1, bf16x4), 0, 1, 2, 3, 4, 5, 6, 7); \
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, ph, o[0], 0, 0, 0);                                       \
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, ph, o[1], 0, 0, 0);                                       \
    } while (0)
#define PP_READ_K(KS, S)                                                                                  \
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                          \
                 : "=&v"(kx##S##0), "=&v"(kx##S##1) : "v"(kaddr##KS), "n"(SK * BUF), "n"(SK * BUF + 32 * ROWB))
#define PP_USE_K(S, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(kx##S##0), "+v"(kx##S##1))
#define PP_QK(KS, S)                                                                                                       \
    do {                                                                                                                   \
        s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx##S##0, qh[KS], (KS) == 0 ? zero : s[0], 0, 0, 0);                \
        s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kx##S##1, qh[KS], (KS) == 0 ? zero : s[1], 0, 0, 0);                \
    } while (0)

    // segment pair u: M segment (slot C = u & 3 holds K(u), slot C - 1 holds V(u - 1), tile u + 2 goes to slot C + 2), then V segment
    auto seg_pair = [&](int u, auto cc) __attribute__((always_inline)) {
        constexpr int C = decltype(cc)::value;
        constexpr int SK = C, SV = (C + 3) & 3, SD = (C + 2) & 3;
        const bool issue = u + 2 < n;
        if (issue) dma_tile((u + 2) * FB_KV, smem + SD * BUF);
        if (wave_active) {
            if (u > 0) {                                   // O += V(u-1)^T . P(u-1)
                fp_s16x4 vxA0, vxA1, vxA2, vxA3, vxB0, vxB1, vxB2, vxB3;
                bf16x8 vf0, vf1;
                PP_READ_V(0, A); PP_READ_V(1, B);
                PP_USE_V(A, 4); PP_PV(0); PP_READ_V(2, A);
                PP_USE_V(B, 4); PP_PV(1); PP_READ_V(3, B);
                PP_USE_V(A, 4); PP_PV(2);
                PP_USE_V(B, 0); PP_PV(3);
            }
            if (u < n) {                                   // S(u) = K(u) . Q^T
                const unsigned kaddr0 = offKs[0], kaddr1 = offKs[1], kaddr2 = offKs[2], kaddr3 = offKs[3];
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                bf16x8 kxA0, kxA1, kxB0, kxB1;
                PP_READ_K(0, A); PP_READ_K(1, B);
                PP_USE_K(A, 2); PP_QK(0, A); PP_READ_K(2, A);
                PP_USE_K(B, 2); PP_QK(1, B); PP_READ_K(3, B);
                PP_USE_K(A, 2); PP_QK(2, A);
                PP_USE_K(B, 0); PP_QK(3, B);
            }
        }
        if (issue) fb_wait_vm<2>(); else fb_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (u >= n) return;
        if (wave_active) {                                 // online softmax of S(u) for this lane's query -> P(u)
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s[0]), "+v"(s[1]));      // MFMA write -> inline-asm VALU read (see the kernel above)
            const int kv0 = u * FB_KV;
            if (kv0 + FB_KV > n_tok) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kv0 + 32 * kb + crow32(r, hi) >= n_tok) s[kb][r] = -INFINITY;
            }
            float mx;
            {
                float ma = max3f(s[0][0], s[0][1], s[0][2]), mb = max3f(s[1][0], s[1][1], s[1][2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) { ma = max3f(ma, s[0][r], s[0][r + 1]); mb = max3f(mb, s[1][r], s[1][r + 1]); }
                mx = max3f(ma, mb, s[0][15]);
                mx = max3f(mx, s[1][15], s[1][15]);
            }
            mx = max3f(mx, __shfl_xor(mx, 32), mx);
            const float m_new = max3f(m_run, mx, mx);       // (key 0 of tile 0 always exists: never -inf)
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            f32x2 rs2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 x = pk_sub(f32x2{s[kb][r], s[kb][r + 1]}, m_new);
                    x[0] = __builtin_amdgcn_exp2f(x[0]);
                    x[1] = __builtin_amdgcn_exp2f(x[1]);
                    rs2 = pk_add(rs2, x);
                    s[kb][r] = x[0];
                    s[kb][r + 1] = x[1];
                }
            float rs = rs2[0] + rs2[1];
            rs += __shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            }
        }
        if (issue) fb_wait_vm<2>(); else fb_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    for (int u = 0; u <= n; u += 4) {
        seg_pair(u, std::integral_constant<int, 0>{});
        if (u + 1 <= n) seg_pair(u + 1, std::integral_constant<int, 1>{});
        if (u + 2 <= n) seg_pair(u + 2, std::integral_constant<int, 2>{});
        if (u + 3 <= n) seg_pair(u + 3, std::integral_constant<int, 3>{});
    }
#undef PP_READ_V
#undef PP_USE_V
#undef PP_PV
#undef PP_READ_K
#undef PP_USE_K
#undef PP_QK
    if (!grp) asm volatile("s_barrier" ::: "memory");

    // ---- normalise, transpose through LDS (wave-private [32 q][68]), coalesced half-row store ----
    __syncthreads();                                    // (the tile buffers are dead: every wave has left its last segment)
    const float inv_l = 1.f / l_run;
    float* so = reinterpret_cast<float*>(smem) + wave * (32 * OPITCH);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) so[li * OPITCH + 32 * b + crow32(r, hi)] = o[b][r] * inv_l;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < D / 8; ++i) {
        const int idx = lane + 64 * i;
        const int r = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
        const int qr = q0 + wave * 32 + r;
        if (qr < n_tok) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(so + r * OPITCH + c4);
            float* orow = O + (size_t)(row_base + qr) * ldo;
            *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(orow) + (col0 + c4) * 2) = __builtin_convertvector(v, bf16x4);
        }
    }
}

