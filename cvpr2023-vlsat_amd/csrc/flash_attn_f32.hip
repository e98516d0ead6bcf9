// Edge cross-attention core, flash style, exact fp32 on the matrix cores.
// Replaces ScaledDotProductAttention.forward's  att = QK^T/sqrt(d); softmax; att.V
// (reference transformer/attention.py:60-76) as called with q = 2D edges, k = v = 3D edges and
// no mask/bias from MMG.forward (reference network_MMG.py:231).  The reference materialises
// att [1,8,E,E] (and a clone); here the E x E scores never leave registers.
//
// Per scene (block-diagonal by construction: one tile entry = one scene), head h:
//   block  = 128 queries (4 waves x 32), loops over the scene's keys in tiles of 32;
//   S^T    = K_tile . Q^T   (32 keys x 32 queries per wave): A operand = K rows from LDS
//            (ds_read_b128, 4 consecutive d per lane), B operand = Q held in 32 VGPRs/lane;
//            "swapped" product so a lane owns ONE query column: its 16 registers are 16 keys,
//            the other 16 keys sit in lane^32 -> the softmax row reductions are 15 in-lane
//            max/add + one cross-half shuffle (no LDS);
//   O^T   += V^T . P^T      : B operand = P straight from the S registers (k index = the key
//            that register already holds), A operand = V[key][d] from LDS (ds_read_b32,
//            conflict free: 32 lanes read 32 consecutive d);
//   online softmax with running (m, l) per query, exp2 with log2(e) folded into the scale.
// K/V tiles are double-buffered in LDS via register staging; one barrier per key tile
// (64 MFMAs = 4096 cycles per wave between barriers).
// Split keys (small problems): one scene alone gives only ceil(T/128)*8 ~ 100 blocks for 256 CUs and a lone block
// runs at ~27 % of a CU's matrix rate, so when a plan has fewer than 512 blocks the key range of every
// (query tile, head) is cut into `parts` pieces, each block writes its un-normalised O plus (m, l) per query, and
// flash_merge_kernel combines them (flash-decoding).  393 -> ~100 us per call at T ~ 2000.
// Roofline: fp32 MFMA.  Algorithmic work 4*T^2*64 flop per (scene, head); HBM traffic is
// Q,O once and K,V re-read once per 128-query block out of L2 (K,V of one scene-head =
// T*64*4*2 B, 0.8 MB at T=1560): the same-scene blocks are made consecutive on one XCD.
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

constexpr int FA_KV = 32;        // keys per tile

// FA_D: head dim = 512 / MODEL.NUM_HEADS (reference network_MMG.py:48-50): 64 as shipped (8 heads), 32 (16 heads) or 128
// (4 heads).  A lane holds FA_D / 2 query values and FA_D / 32 output accumulators; everything else is the same kernel.
template <int FA_D>
__global__ __launch_bounds__(256, 2) void flash_attn_f32_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, int ldq, int ldkv, int ldo, const int4* __restrict__ tiles, int n_tiles,
    float scale_log2e, FlashSplit sp) {
    constexpr int FA_PITCH = FA_D + 4;   // LDS row pitch (floats): 16-B pad -> conflict-free b128 reads
    constexpr int HD = FA_D / 2, NO = FA_D / 32, NS = FA_D / 32;     // query values per lane, output blocks, staged float4 per thread and operand
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * FA_KV * FA_PITCH];   // [buf][K|V][32][FA_D + 4]
    constexpr int BUF = 2 * FA_KV * FA_PITCH;

    const int tile_id = xcd_remap(blockIdx.x, n_tiles);
    const int4 t = tiles[tile_id];
    const int row_base = t.x, n_tok = t.y, q0 = t.z, head = t.w;
    // split mode: sp.krange[tile] = {first key tile, end key tile, part, -}
    const int4 kr = sp.parts > 1 ? sp.krange[tile_id] : make_int4(0, (n_tok + FA_KV - 1) / FA_KV, 0, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const size_t col0 = (size_t)head * FA_D;

    // ---- this lane's query row: d = HD*hi + s, s = 0..HD-1 (pre-scaled) ----
    // A wave whose 32 query rows are all past the scene's token count still helps staging K/V
    // but issues no MFMA (the last 128-row block of a scene is usually mostly empty:
    // 1560 = 12*128 + 24), leaving the matrix pipe to the other resident blocks.
    const bool wave_active = q0 + wave * 32 < n_tok;
    int qrow = q0 + wave * 32 + li;
    if (qrow >= n_tok) qrow = n_tok - 1;      // clamped rows are computed but never stored
    float q[HD];
    {
        const float* qp = Q + (size_t)(row_base + qrow) * ldq + col0 + HD * hi;
#pragma unroll
        for (int g = 0; g < HD / 4; ++g) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(qp + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) q[4 * g + c] = x[c] * scale_log2e;
        }
    }

    f32x16 o[NO];
#pragma unroll
    for (int b = 0; b < NO; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: K and V tile each [32][FA_D] = 8 FA_D float4; thread -> column chunk tid % C4 of rows tid / C4 + RS i
    constexpr int C4 = FA_D / 4, RS = 256 / C4;  // float4 per row; rows covered by one pass of the 256 threads
    const int srow = tid / C4, sc4 = (tid % C4) * 4;
    f32x4 rk[NS], rv[NS];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            int r = kv0 + srow + RS * i;
            r = r < n_tok ? r : n_tok - 1;
            const size_t off = (size_t)(row_base + r) * ldkv + col0 + sc4;
            rk[i] = *reinterpret_cast<const f32x4*>(K + off);
            rv[i] = *reinterpret_cast<const f32x4*>(V + off);
        }
    };
    auto store_tile = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            *reinterpret_cast<f32x4*>(buf + (srow + RS * i) * FA_PITCH + sc4) = rk[i];
            *reinterpret_cast<f32x4*>(buf + FA_KV * FA_PITCH + (srow + RS * i) * FA_PITCH + sc4) = rv[i];
        }
    };

    const int n_kv_tiles = (n_tok + FA_KV - 1) / FA_KV;
    const int kt0 = kr.x, kt1 = kr.y;
    if (kt0 < kt1) {
        load_tile(kt0 * FA_KV);
        store_tile(smem + (kt0 & 1) * BUF);
    }
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const float* sK = smem + (kt & 1) * BUF;
        const float* sV = sK + FA_KV * FA_PITCH;
        const bool more = kt + 1 < kt1;
        if (more && !(kExperiments && (sp.ablate & 1))) load_tile((kt + 1) * FA_KV);

        if (wave_active) {
        // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d] ----
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < HD / 4; ++g) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(sK + li * FA_PITCH + HD * hi + 4 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], q[4 * g + c], s, 0, 0, 0);
        }
        // keys beyond the scene's token count (last tile only)
        if (kt == n_kv_tiles - 1) {
            const int kv0 = kt * FA_KV;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kv0 + crow32(r, hi) >= n_tok) s[r] = -INFINITY;
        }
        // ---- online softmax for this lane's query ----
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = half_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
            rs += s[r];
        }
        rs = half_sum(rs);
        l_run = l_run * alpha + rs;
        m_run = m_new;
        // rescale the running output only when some query's maximum moved (alpha == 1 exactly otherwise: after the
        // first few key tiles this skips 32 multiplies per tile for the whole wave without changing a bit)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int b = 0; b < NO; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
        }
        // ---- O^T[d][query] += sum_key V[key][d] * P[key][query] ----
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vp = sV + crow32(r, hi) * FA_PITCH + li;
#pragma unroll
            for (int b = 0; b < NO; ++b) o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * b], s[r], o[b], 0, 0, 0);
        }
        }   // wave_active
        if (more && !(kExperiments && (sp.ablate & 2))) store_tile(smem + ((kt + 1) & 1) * BUF);
        __syncthreads();
    }

    // ---- normalise (or, in split mode, keep un-normalised and record m, l), transpose through LDS
    //      (wave-private [32 q][68]), coalesced store ----
    const bool split = sp.parts > 1;
    const float inv_l = split ? 1.f : 1.f / l_run;
    if (split) {
        O = sp.o_part + (size_t)kr.z * sp.part_stride;
        const int qr = q0 + wave * 32 + li;
        if (hi == 0 && qr < n_tok) {
            const size_t i = ((size_t)kr.z * sp.rows + row_base + qr) * sp.heads + head;
            sp.m_part[i] = m_run;
            sp.l_part[i] = l_run;
        }
    }
    float* so = smem + wave * (32 * FA_PITCH);     // 4 x 32 rows = all of smem
#pragma unroll
    for (int b = 0; b < NO; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) so[li * FA_PITCH + 32 * b + crow32(r, hi)] = o[b][r] * inv_l;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FA_D / 8; ++i) {
        const int idx = lane + 64 * i;             // 8 FA_D float4 = 32 rows x FA_D / 4
        const int r = idx / C4, c4 = (idx % C4) * 4;
        const int qr = q0 + wave * 32 + r;
        if (qr < n_tok)
            *reinterpret_cast<f32x4*>(O + (size_t)(row_base + qr) * ldo + col0 + c4) =
                *reinterpret_cast<const f32x4*>(so + r * FA_PITCH + c4);
    }
}

// O[row, h*64 + d] = sum_p 2^(m_p - m) O_p[row, h*64 + d] / sum_p 2^(m_p - m) l_p,  m = max_p m_p  (per row, head)
__global__ __launch_bounds__(256) void flash_merge_kernel(float* __restrict__ O, int ldo, FlashSplit sp, int out_split, int FA_D) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;       // one float4 of one row
    const int per_row = sp.heads * (FA_D / 4);
    const size_t row = idx / per_row;
    if (row >= (size_t)sp.rows) return;
    const int c4 = (int)(idx % per_row), head = c4 / (FA_D / 4);
    float m = -INFINITY;
    for (int p = 0; p < sp.parts; ++p) m = fmaxf(m, sp.m_part[((size_t)p * sp.rows + row) * sp.heads + head]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int p = 0; p < sp.parts; ++p) {
        const size_t i = ((size_t)p * sp.rows + row) * sp.heads + head;
        const float w = __builtin_amdgcn_exp2f(sp.m_part[i] - m);
        if (w > 0.f) {                                   // an empty part has m = -inf: its O slot was never written
            den += w * sp.l_part[i];
            const f32x4 o = *reinterpret_cast<const f32x4*>(sp.o_part + (size_t)p * sp.part_stride + row * ldo + c4 * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += w * o[c];
        }
    }
    const float inv = 1.f / den;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        acc[c] *= inv;
        if (out_split == 1) acc[c] = pack_split(acc[c]);
    }
    if (out_split == 3) {                               // half rows of fp16
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_fmed3f(acc[c], -65504.f, 65504.f);
        *reinterpret_cast<f16x4_t*>(reinterpret_cast<char*>(O + row * ldo) + c4 * 8) = __builtin_convertvector(acc, f16x4_t);
        return;
    }
    if (out_split == 2) {                               // half rows
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<char*>(O + row * ldo) + c4 * 8) = __builtin_convertvector(acc, bf16x4_t);
        return;
    }
    *reinterpret_cast<f32x4*>(O + row * ldo + c4 * 4) = acc;
}

int launch_flash_attn(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo,
                      const int4* tiles, int n_tiles, float scale_log2e, hipStream_t s, const FlashSplit* split, int head_dim) {
    if (n_tiles <= 0) return 0;
    const int FA_D = head_dim;
    if (FA_D != 32 && FA_D != 64 && FA_D != 128) return fail(-1, "flash_attn: head dim must be 32, 64 or 128");
    if ((ldq | ldkv | ldo) & 3) return fail(-1, "flash_attn: leading dims must be multiples of 4");
    FlashSplit sp{};
    if (split && split->parts > 1) {
        sp = *split;
        if (!sp.krange || !sp.o_part || !sp.m_part || !sp.l_part || sp.heads * FA_D > ldo)
            return fail(-1, "flash_attn: incomplete split-key workspace");
    }
    if (split) sp.ablate = split->ablate;
    if (FA_D == 32)
        hipLaunchKernelGGL(flash_attn_f32_kernel<32>, dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
    else if (FA_D == 64)
        hipLaunchKernelGGL(flash_attn_f32_kernel<64>, dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
    else
        hipLaunchKernelGGL(flash_attn_f32_kernel<128>, dim3(n_tiles), dim3(256), 0, s, Q, K, V, O, ldq, ldkv, ldo, tiles, n_tiles, scale_log2e, sp);
    VLSAT_LAUNCH_CHECK("flash_attn_f32");
    if (sp.parts > 1) return launch_flash_merge(O, ldo, sp, s, 0, FA_D);
    return 0;
}

int launch_flash_merge(float* O, int ldo, const FlashSplit& sp, hipStream_t s, int out_split, int head_dim) {
    const int FA_D = head_dim;
    const size_t n4 = (size_t)sp.rows * sp.heads * (FA_D / 4);
    hipLaunchKernelGGL(flash_merge_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, O, ldo, sp, out_split, FA_D);
    VLSAT_LAUNCH_CHECK("flash_merge");
    return 0;
}

}  // namespace vlsat
