// PointNet object encoder on the bf16 matrix cores (BASELINE configs[2]): the same fusion as pointnet.hip --
//   out[n, :] = max_p relu(W3 relu(W2 relu(W1 x_p + b1) + b2) + b3),  reference network_PointNet.py:141-164 --
// with conv2 / conv3 on v_mfma_f32_32x32x16_bf16.  TERMS = 3: activations and weights as bf16 hi + lo, three MFMAs
// per product (~1e-5); TERMS = 1: single-rounded operands.
//
// One block = 8 waves (4 x 2) walks 128-point chunks of one object (twice the rows of the fp32 kernel: with 16x faster
// matrix instructions the W3 stream from L2 -- 6 column chunks x 4 k-slices per point chunk -- would otherwise bound it):
//   conv1 (CIN->64)   VALU; the result goes to LDS as bf16 planes H1[128][64] (row pitch 144 B)
//   conv2 (64->128)   A = H1 planes, B = W2 planes [128][64] staged once per point chunk; relu(.+b2) -> H2 planes
//                     [128][128] (pitch 272 B), written over H1
//   conv3 (128->768)  W3 plane slices [128 cols][32 k] (unpadded 64-byte rows, 16-byte pieces XOR-swizzled with bits 2..3 of
//                     the row: conflict-free for the staging stores AND the fragment reads) double-buffered through
//                     registers; epilogue relu + running column max in registers
// The H / W2 pitches are 16 B off a multiple of 128 B: conflict-free ds_read_b128 fragment reads.
// The bf16 planes of W2 / W3 are the ones vlsat_set_gemm_precision makes for every GEMM weight.
#include "gemm_core.h"
#include "kernels.h"

namespace vlsat {

namespace {

constexpr int PB_M = 128;                 // points per chunk
constexpr int PB_P1 = 144, PB_P2 = 272;   // row pitch (bytes) of the H1 / H2 planes
constexpr int PB_PW2 = 144, PB_PW3 = 64;  // row pitch of the W2 planes [128][64] / W3 slices [128][32] (unpadded, pieces XOR-swizzled)
constexpr int PB_H = PB_M * PB_P2;        // bytes of one H plane region (H2 is the larger)
constexpr int PB_W = 128 * PB_PW2;        // bytes of one W2 plane
constexpr int PB_W3 = 128 * PB_PW3;       // bytes of one W3 slice plane
constexpr int PB_MAXNC = 6;

template <int PL, bool F16 = false>
__device__ __forceinline__ void store8(char* plane0, int plane_stride, int off, const f32x4& a, const f32x4& b) {
    if constexpr (F16) {                     // (TERMS = 2: single-rounded fp16 operands, precision mode fp16_mixed)
        typedef _Float16 f16x4_s __attribute__((ext_vector_type(4)));
        f32x4 ca, cb;
#pragma unroll
        for (int c = 0; c < 4; ++c) { ca[c] = __builtin_amdgcn_fmed3f(a[c], -65504.f, 65504.f); cb[c] = __builtin_amdgcn_fmed3f(b[c], -65504.f, 65504.f); }
        const bf16x4 g0 = __builtin_bit_cast(bf16x4, __builtin_convertvector(ca, f16x4_s)), g1 = __builtin_bit_cast(bf16x4, __builtin_convertvector(cb, f16x4_s));
        *reinterpret_cast<bf16x8*>(plane0 + off) = __builtin_shufflevector(g0, g1, 0, 1, 2, 3, 4, 5, 6, 7);
        return;
    }
    const bf16x4 h0 = __builtin_convertvector(a, bf16x4), h1 = __builtin_convertvector(b, bf16x4);
    *reinterpret_cast<bf16x8*>(plane0 + off) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    if (PL == 2) {
        const bf16x4 l0 = __builtin_convertvector(a - __builtin_convertvector(h0, f32x4), bf16x4);
        const bf16x4 l1 = __builtin_convertvector(b - __builtin_convertvector(h1, f32x4), bf16x4);
        *reinterpret_cast<bf16x8*>(plane0 + plane_stride + off) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <int CIN, int TERMS>
__global__ __launch_bounds__(512, 2) void pointnet_bf16_kernel(
    const float* __restrict__ pts, int P, const float* __restrict__ w1, const float* __restrict__ b1,
    const uint16_t* __restrict__ w2h, const uint16_t* __restrict__ w2l, const float* __restrict__ b2,
    const uint16_t* __restrict__ w3h, const uint16_t* __restrict__ w3l, const float* __restrict__ b3, int n_out,
    float* __restrict__ out, int nsplit) {
    constexpr int PL = TERMS == 3 ? 2 : 1;
    constexpr bool F16 = TERMS == 2;             // fp16 operands (w2h / w3h are then fp16 planes)
    constexpr int S1 = (CIN + 4) & ~3;
    constexpr int WREG = 2 * PL * PB_W3 > PL * PB_W ? 2 * PL * PB_W3 : PL * PB_W;      // conv2: [PL][128][144]; conv3: 2 stages of [PL][128][80]
    __shared__ __attribute__((aligned(16))) char smem[PL * PB_H + WREG + 64 * S1 * 4];
    char* sH = smem;                                   // [PL][128][pitch]
    char* sW = smem + PL * PB_H;
    float* sW1 = reinterpret_cast<float*>(smem + PL * PB_H + WREG);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hi = lane >> 5;
    const int obj = blockIdx.x / nsplit, part = blockIdx.x % nsplit;
    const int n_chunks = (P + PB_M - 1) / PB_M;
    const int n_nc = n_out / 128;
    const float* op = pts + (size_t)obj * CIN * P;

    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) sW1[tid * S1 + c] = w1[tid * CIN + c];
        sW1[tid * S1 + CIN] = b1[tid];
    }
    float rmax[PB_MAXNC][2];
#pragma unroll
    for (int i = 0; i < PB_MAXNC; ++i) rmax[i][0] = rmax[i][1] = 0.f;
    __syncthreads();

    // W staging: a plane slice of 128 rows; thread -> (row, 16-byte piece)
    const uint16_t* wsrc[2] = {w3h, w3l};
    uint4 rw[PL];
    auto w3_load = [&](int j) {                 // slice j = (column chunk j >> 2, k-slice j & 3): rows 64 B, 4 pieces -> 512 pieces
        const int row = tid >> 2, piece = tid & 3;
        int n = (j >> 2) * 128 + row;
        n = n < n_out ? n : n_out - 1;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
            rw[pl] = *reinterpret_cast<const uint4*>(wsrc[pl] + (size_t)n * 128 + (j & 3) * 32 + piece * 8);
    };
    auto w3_store = [&](int stage) {            // unpadded 64-byte rows, piece ^ ((row >> 2) & 3): 16 lanes = 4 rows x 4 pieces = 16 bank groups
        const int row = tid >> 2, piece = (tid & 3) ^ ((tid >> 4) & 3);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl)
            *reinterpret_cast<uint4*>(sW + (stage * PL + pl) * (128 * PB_PW3) + row * PB_PW3 + piece * 16) = rw[pl];
    };

    for (int ch = part; ch < n_chunks; ch += nsplit) {
        // ---- conv1: thread -> point pp, channels cg .. cg+15 ----
        {
            // 16 consecutive lanes (one pass of a ds_write_b128) = 16 points at the same channel group: with the 144-byte pitch
            // their bank groups 9 pp (mod 16) are a permutation (tid >> 2, tid & 3 wrapped onto the first point's banks from the
            // third point on), and the point reads are coalesced
            const int pp = wave * 16 + (lane & 15), cg = (lane >> 4) * 16;
            int p = ch * PB_M + pp;
            p = p < P ? p : P - 1;
            float xin[CIN];
#pragma unroll
            for (int c = 0; c < CIN; ++c) xin[c] = op[(size_t)c * P + p];
            f32x4 hq[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* w = sW1 + (cg + c4 * 4 + c) * S1;
                    float a = w[CIN];
#pragma unroll
                    for (int k = CIN - 1; k >= 0; --k) a = fmaf(w[k], xin[k], a);
                    hq[c4][c] = fmaxf(a, 0.f);
                }
            store8<PL, F16>(sH, PB_H, pp * PB_P1 + cg * 2, hq[0], hq[1]);
            store8<PL, F16>(sH, PB_H, pp * PB_P1 + cg * 2 + 16, hq[2], hq[3]);
        }
        // ---- W2 planes [128][64] -> LDS (pitch 144): 128 rows x 8 pieces = 1024 pieces per plane ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i, row = idx >> 3, piece = idx & 7;
            *reinterpret_cast<uint4*>(sW + row * PB_PW2 + piece * 16) = *reinterpret_cast<const uint4*>(w2h + row * 64 + piece * 8);
            if (PL == 2)
                *reinterpret_cast<uint4*>(sW + PB_W + row * PB_PW2 + piece * 16) = *reinterpret_cast<const uint4*>(w2l + row * 64 + piece * 8);
        }
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        {
            const char* ap = sH + (wm * 32 + li) * PB_P1 + 16 * hi;
            const char* bp = sW + (wn * 64 + li) * PB_PW2 + 16 * hi;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + 32 * ks);
                bf16x8 al = ah;
                if (PL == 2) al = *reinterpret_cast<const bf16x8*>(ap + PB_H + 32 * ks);
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp + tn * 32 * PB_PW2 + 32 * ks);
                    if (PL == 2) {
                        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + PB_W + tn * 32 * PB_PW2 + 32 * ks);
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[tn], 0, 0, 0);
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[tn], 0, 0, 0);
                    }
                    acc[tn] = mfma_h<F16>(ah, bh, acc[tn]);
                }
            }
        }
        w3_load(0);                                    // first W3 slice in flight over the barrier
        __syncthreads();                               // everyone done reading H1 / W2
        // H2[point][col] = relu(acc + b2) as bf16 planes; lane: column (li), rows crow32(r, hi)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int col = wn * 64 + tn * 32 + li;
            const float bb = b2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaxf(acc[tn][r] + bb, 0.f);
                char* dst = sH + (wm * 32 + crow32(r, hi)) * PB_P2 + col * 2;
                if constexpr (F16) { *reinterpret_cast<_Float16*>(dst) = (_Float16)fminf(v, 65504.f); continue; }
                const __bf16 h = (__bf16)v;
                *reinterpret_cast<__bf16*>(dst) = h;
                if (PL == 2) *reinterpret_cast<__bf16*>(dst + PB_H) = (__bf16)(v - (float)h);
            }
        }
        w3_store(0);
        __syncthreads();
        // ---- conv3: n_nc column chunks x 4 k-slices of 32 ----
        const int n_slices = n_nc * 4;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        for (int j = 0; j < n_slices; ++j) {
            const int nc = j >> 2, k4 = j & 3;
            const bool more = j + 1 < n_slices;
            if (more) w3_load(j + 1);
            const char* ap = sH + (wm * 32 + li) * PB_P2 + 64 * k4 + 16 * hi;
            const char* bp = sW + ((j & 1) * PL) * (128 * PB_PW3) + (wn * 64 + li) * PB_PW3;
            const int sw = (li >> 2) & 3;                             // piece swizzle of this lane's rows (w3_store)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap + 32 * ks);
                bf16x8 al = ah;
                if (PL == 2) al = *reinterpret_cast<const bf16x8*>(ap + PB_H + 32 * ks);
                const int po = ((2 * ks + hi) ^ sw) * 16;
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(bp + tn * 32 * PB_PW3 + po);
                    if (PL == 2) {
                        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(bp + 128 * PB_PW3 + tn * 32 * PB_PW3 + po);
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[tn], 0, 0, 0);
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[tn], 0, 0, 0);
                    }
                    acc[tn] = mfma_h<F16>(ah, bh, acc[tn]);
                }
            }
            if (k4 == 3) {
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const float bb = b3[nc * 128 + wn * 64 + tn * 32 + li];
                    float m = 0.f;                                   // relu folded into the max with 0
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[tn][r] + bb);
                    m = half_max(m);
#pragma unroll
                    for (int q = 0; q < PB_MAXNC; ++q)
                        if (q == nc) rmax[q][tn] = fmaxf(rmax[q][tn], m);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
                }
            }
            if (more) w3_store((j + 1) & 1);
            __syncthreads();
        }
    }
    // ---- merge the four row groups (wm) and write / atomically merge ----
    float* sR = reinterpret_cast<float*>(sH);                          // [4][768]
    if (hi == 0) {
#pragma unroll
        for (int q = 0; q < PB_MAXNC; ++q)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                if (q < n_nc) sR[wm * 768 + q * 128 + wn * 64 + tn * 32 + li] = rmax[q][tn];
    }
    __syncthreads();
    for (int c = tid; c < n_out; c += 512) {
        const float v = fmaxf(fmaxf(sR[c], sR[768 + c]), fmaxf(sR[2 * 768 + c], sR[3 * 768 + c]));
        float* dst = out + (size_t)obj * n_out + c;
        if (nsplit == 1) *dst = v;
        else atomicMax(reinterpret_cast<int*>(dst), __float_as_int(v));
    }
}

}  // namespace

int launch_pointnet_bf16(const float* pts, int n_obj, int n_points, int cin, const float* w1, const float* b1,
                         const uint16_t* w2h, const uint16_t* w2l, const float* b2, const uint16_t* w3h, const uint16_t* w3l,
                         const float* b3, int n_out, int terms, float* out, hipStream_t s) {
    if (n_obj <= 0) return 0;
    if (n_points <= 0) return fail(-1, "pointnet: n_points must be > 0");
    if (n_out % 128 || n_out > 128 * PB_MAXNC) return fail(-1, "pointnet: n_out must be a multiple of 128, <= 768");
    if (terms != 1 && terms != 2 && terms != 3) return fail(-1, "pointnet_bf16: terms must be 1, 3 or 2 (single-rounded fp16)");
    const int n_chunks = (n_points + PB_M - 1) / PB_M;
    int nsplit = (256 + n_obj - 1) / n_obj;
    if (nsplit > n_chunks) nsplit = n_chunks;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 1 && launch_zero_f32(out, (size_t)n_obj * n_out, s)) return -1;
#define VLSAT_PB(CIN, T) hipLaunchKernelGGL((pointnet_bf16_kernel<CIN, T>), dim3(n_obj * nsplit), dim3(512), 0, s, pts, n_points, w1, b1, \
                                            w2h, w2l, b2, w3h, w3l, b3, n_out, out, nsplit)
#define VLSAT_PB_CASE(CIN) case CIN: if (terms == 3) VLSAT_PB(CIN, 3); else if (terms == 2) VLSAT_PB(CIN, 2); else VLSAT_PB(CIN, 1); break;
    switch (cin) {
        VLSAT_PB_CASE(3) VLSAT_PB_CASE(6) VLSAT_PB_CASE(9)
        default: return fail(-1, "pointnet: point channels must be 3, 6 or 9");
    }
#undef VLSAT_PB_CASE
#undef VLSAT_PB
    VLSAT_LAUNCH_CHECK("pointnet_bf16");
    return 0;
}

}  // namespace vlsat
