// Kernel-level C entry points (unit tests and tools drive single kernels through these) and debug hooks.
#include <cstring>
#include <vector>

#include "engine.h"

using namespace vlsat;

extern "C" {

const char* vlsat_last_error(void) { return last_error_cstr(); }

// debug: stop the forward after stage `stage` (see DESIGN.md "debug stages"); -1 = run all
int vlsat_debug_stop_after(vlsat_handle h, int32_t stage) {
    if (!h) return fail(VLSAT_EINVAL, "null handle");
    h->debug_stop = stage;
    return 0;
}

// debug: while `buf` (device, >= 4 * 512 int64) is set, every persistent GEMM block writes
// {shader cycles, 100 MHz wall ticks, tiles done, 1} at exit: effective clock = cycles / (ticks / 1e8)
int vlsat_debug_gemm_clock_probe(int64_t* buf) {
    gemm_set_clock_probe(reinterpret_cast<long long*>(buf));
    return 0;
}

// Keys of the edge cross-attention for plans created from now on (see vlsat.h)
int vlsat_set_edge_attention_scope(vlsat_handle h, int32_t scope) {
    if (!h) return fail(VLSAT_EINVAL, "null handle");
    if (scope != 0 && scope != 1) return fail(VLSAT_EINVAL, "edge attention scope: 0 (per scene) or 1 (whole batch)");
    h->edge_scope = scope;
    ++h->config_epoch;
    return 0;
}

// debug / experiment switches of one handle (they replace the VLSAT_* environment variables of round 1; defaults are
// the measured-best settings and none changes results beyond fp32 summation order):
//   "dual_stream" 0|1|2 2D twin stages on a second stream: 1 small plans only, 2 every plan (plans created afterwards)
//   "flash_split" 0|1   split-key edge attention for plans that cannot fill the chip (plans created afterwards)
//   "gemm_dma"    0|1   LDS-direct staging of fp32 GEMM operands (0: VGPR-staged)
//   "gemm_p8"     0|1   single-rounding bf16 modes: large half-row launches on the 256 x 256 8-phase kernel (0: ring kernel)
//   "gate_grid"   n     persistent grid of the gate kernel (0: default)
//   "prof_dual"   0|1   per-class profiling keeps the two-stream execution (1, default) or serialises on the launch stream
//   "gate_heads_bf16" 0|1   bf16 modes at other head geometries: the gate on the bf16 kernel (1, default) or the fp32 one
//   "flash_heads_bf16" 0|1  bf16 modes at 4 / 16 heads: edge attention on the bf16 kernel (1, default) or the fp32 one
//   "gate_heads_mfma" 0|1  non-default head geometries: the MFMA gate kernel (1, default) or the VALU one
//   "gate_row_map" 0|1  rows of a gate wave: 32 edges of one head (1, default) or 4 edges x 8 heads (0)
//   "node_attn_split" n  node attention with sixteen lanes per query for plans of fewer than n one-query-per-lane waves
//   "gemm_splitk" 0|1   small GEMM launches on the split-K kernel (0: everything on the persistent kernel)
//   "split_fmt"   0|1   bf16 modes: edge tensors between matrix kernels as bf16 hi/lo pairs (0: plain fp32, split on read)
//   "ln_resid"    0|1   split-bf16 mode: edge-attention residual added by the LayerNorm kernel (0: in the out-projection GEMM)
//   "half_fmt"    0|1   single-rounding modes: those tensors as plain bf16 at half the traffic (0: split pairs)
//   "flash_bf16"  0|1   bf16 modes: edge attention on the bf16 matrix cores (0: keep the fp32 kernel)
//   "pointnet_bf16", "gate_bf16" 0|1   bf16 modes: object encoder / edge gate on the bf16 matrix cores (0: fp32 kernels)
//   "flash_tr"    0|1   bf16 attention: V operand by ds_read_b64_tr_b16 (0: ds_read_u16 gather)
int vlsat_debug_option(vlsat_handle h, const char* name, int32_t value) {
    if (!h || !name) return fail(VLSAT_EINVAL, "vlsat_debug_option: null argument");
    const std::string k(name);
    ++h->config_epoch;
    if (k == "dual_stream") h->dual_stream = value < 0 ? 0 : value;
    else if (k == "flash_split") h->fa_split = value != 0;
    else if (k == "gemm_dma") h->gemm_no_dma = value == 0;
    else if (k == "gemm_p8") h->gemm_no_p8 = value == 0;
    else if (k == "gate_grid") h->gate_grid = value > 0 ? value : 0;
    else if (k == "gate_row_map") h->gate_row_map = value != 0;
    else if (k == "prof_dual") h->prof_dual = value != 0;
    else if (k == "gate_heads_bf16") h->gate_heads_bf16 = value != 0;
    else if (k == "flash_heads_bf16") h->flash_heads_bf16 = value != 0;
    else if (k == "gate_heads_mfma") h->gate_heads_mfma = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (k == "gemm_splitk") h->gemm_splitk = value != 0;
    else if (k == "node_attn_split") h->node_attn_split = value;
    else if (k == "split_fmt") h->split_fmt = value != 0;
    else if (k == "half_fmt") h->half_fmt = value != 0;
    else if (k == "ln_resid") h->ln_resid = value != 0;
    else if (k == "flash_pv_terms") h->flash_pv_terms = value == 2 ? 2 : 3;
    else if (k == "flash_bf16") h->flash_bf16 = value != 0;
    else if (k == "pointnet_bf16") h->pointnet_bf16 = value != 0;
    else if (k == "gate_bf16") h->gate_bf16 = value != 0;
    else if (k == "flash_tr") h->flash_tr = value != 0;
    else return fail(VLSAT_EINVAL, "vlsat_debug_option: unknown option " + k);
    return 0;
}

// -------------------------------------------------------------------------------------------
// workspace of the split-K kernel for the handle-free test entry points (allocated on first use, never freed)
static int test_splitk_ws(GemmArgs& a) {
    static float* ws = nullptr;
    static unsigned* cnt = nullptr;
    if (!ws) {
        VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ws), SPLITK_WS_FLOATS * sizeof(float)));
        VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&cnt), SPLITK_COUNTERS * sizeof(unsigned)));
        VLSAT_HIP_CHECK(hipMemset(cnt, 0, SPLITK_COUNTERS * sizeof(unsigned)));
    }
    a.sk_ws = ws; a.sk_ws_floats = SPLITK_WS_FLOATS; a.sk_counters = cnt; a.sk_n_counters = SPLITK_COUNTERS;
    return 0;
}

int vlsat_k_gemm(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc, int32_t M, int32_t N,
                 int32_t K, const float* bias, const float* rowscale, const float* resid, int32_t ldr, float resid_scale,
                 const float* g0, const int32_t* gi0, int32_t ldg0, const float* g1, const int32_t* gi1, int32_t ldg1,
                 int32_t relu_a, int32_t act, void* stream) {
    GemmArgs a;
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.bias = bias; a.rowscale = rowscale; a.resid = resid; a.ldr = ldr; a.resid_scale = resid_scale;
    a.g0 = g0; a.gi0 = gi0; a.ldg0 = ldg0; a.g1 = g1; a.gi1 = gi1; a.ldg1 = ldg1; a.relu_a = relu_a & 1; a.act = act;
    if (relu_a & 2) a.prefetch = 0;             // (bit 1 of relu_a: no A-panel prefetch -- benchmarking)
    if (relu_a & 4) RUN(test_splitk_ws(a));     // (bit 2: small launches may take the split-K kernel)
    if (relu_a & 8) a.no_p8 = 1;                // (bit 3: large launches stay off the 256 x 256 8-phase kernel)
    return launch_gemm(a, static_cast<hipStream_t>(stream));
}

// w[i] -> bf16 hi[i] + bf16 lo[i]: the planes the bf16 GEMM modes read (asynchronous on `stream`)
int vlsat_k_split_bf16(const float* w, size_t n, uint16_t* hi, uint16_t* lo, void* stream) {
    if (!w || !hi || !lo) return fail(VLSAT_EINVAL, "split_bf16: null argument");
    return launch_split_bf16(w, n, hi, lo, static_cast<hipStream_t>(stream));
}

// vlsat_k_gemm on the bf16 matrix cores with caller-provided weight planes (asynchronous): prec 1 = bf16, 3 = split-bf16;
// no_dma = 1 selects the VGPR-staged operand pipe of round 1, prefetch = slices of look-ahead of the A prefetch (0 = off)
int vlsat_k_gemm_planes(const float* A, int32_t lda, const float* W, const uint16_t* Whi, const uint16_t* Wlo, int32_t ldw,
                        float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias,
                        const float* resid, int32_t ldr, float resid_scale,
                        const float* g0, const int32_t* gi0, int32_t ldg0, const float* g1, const int32_t* gi1, int32_t ldg1,
                        int32_t relu_a, int32_t act, int32_t prec, int32_t no_dma, int32_t prefetch, int32_t fmt, float c_scale,
                        void* stream) {
    if (prec != 1 && prec != 3) return fail(VLSAT_EINVAL, "gemm_planes: prec must be 1 or 3");
    GemmArgs a;
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.bias = bias; a.resid = resid; a.ldr = ldr; a.resid_scale = resid_scale;
    a.g0 = g0; a.gi0 = gi0; a.ldg0 = ldg0; a.g1 = g1; a.gi1 = gi1; a.ldg1 = ldg1; a.relu_a = relu_a; a.act = act;
    a.prec = prec; a.Whi = Whi; a.Wlo = Wlo; a.no_dma = no_dma; a.prefetch = prefetch;
    const int code = (fmt >> 5) & 1 ? 2 : 1;   // (bit 5: the flagged operands are half rows instead of split pairs)
    a.a_split = (fmt & 1) * code; a.r_split = ((fmt >> 1) & 1) * code; a.c_split = ((fmt >> 2) & 1) * code; a.c_scale = c_scale;
    // (bit 3 was the k-rotation experiment: removed; bit 4: no ring kernel -- benchmarking)
    a.no_ring = (fmt >> 4) & 1;
    a.ring_wide = (fmt >> 7) & 1;              // (bit 7: ring kernel with 128 x 256 tiles where N allows)
    a.ablate = ((fmt >> 8) & 3) | (((fmt >> 13) & 63) << 2);   // (bits 8, 9, 13..18: timing experiments, see GemmArgs::ablate)
    a.ring_bk32 = (fmt >> 10) & 1;             // (bit 10: half-row ring kernel with 32-wide k slices)
    a.ring_nodb = (fmt >> 11) & 1;             // (bit 11: ... without the double-buffered fragment sets)
    a.no_p8 = (fmt >> 12) & 1;                 // (bit 12: half-row launches skip the 256 x 256 8-phase kernel)
    if ((fmt >> 6) & 1) RUN(test_splitk_ws(a));    // (bit 6: small launches may take the split-K kernel)
    return launch_gemm(a, static_cast<hipStream_t>(stream));
}

// test entry point: splits a dense W [N,K] on the spot (allocates, synchronises) and runs vlsat_k_gemm_planes
int vlsat_k_gemm_bf16(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc, int32_t M, int32_t N,
                      int32_t K, const float* bias, const float* resid, int32_t ldr, float resid_scale,
                      const float* g0, const int32_t* gi0, int32_t ldg0, const float* g1, const int32_t* gi1, int32_t ldg1,
                      int32_t relu_a, int32_t act, int32_t prec, int32_t no_dma, void* stream) {
    if (ldw != K) return fail(VLSAT_EINVAL, "gemm_bf16: W must be dense [N,K] (the planes are made from it)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)N * K;
    uint16_t *hi = nullptr, *lo = nullptr;
    VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&hi), n * 2 + 256));
    VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&lo), n * 2 + 256));
    int r = launch_split_bf16(W, n, hi, lo, st);
    if (!r) r = vlsat_k_gemm_planes(A, lda, W, hi, lo, ldw, C, ldc, M, N, K, bias, resid, ldr, resid_scale, g0, gi0, ldg0, g1, gi1,
                                    ldg1, relu_a, act, prec, no_dma, -1, 0, 1.f, stream);
    hipStreamSynchronize(st);
    hipFree(hi);
    hipFree(lo);
    return r;
}

int vlsat_k_pointnet(const float* pts, int32_t n_obj, int32_t n_points, const float* w1, const float* b1,
                     const float* w2, const float* b2, const float* w3, const float* b3, int32_t n_out, float* out,
                     void* stream) {
    return launch_pointnet(pts, n_obj, n_points, 3, w1, b1, w2, b2, w3, b3, n_out, out, static_cast<hipStream_t>(stream));
}

int vlsat_k_flash_attn(const float* Q, const float* K, const float* V, float* O, int32_t ld, const int64_t* tok_ptr,
                       int32_t n_scenes, int32_t n_heads, float scale, void* stream) {
    if (!tok_ptr || n_scenes <= 0) return fail(VLSAT_EINVAL, "flash_attn: bad scene table");
    std::vector<int4> tiles;
    for (int s = 0; s < n_scenes; ++s) {
        const int64_t T = tok_ptr[s + 1] - tok_ptr[s];
        for (int hh = 0; hh < n_heads; ++hh)
            for (int64_t q0 = 0; q0 < T; q0 += FLASH_BQ) tiles.push_back(make_int4((int)tok_ptr[s], (int)T, (int)q0, hh));
    }
    if (tiles.empty()) return 0;
    int4* d = nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), tiles.size() * sizeof(int4)));
    VLSAT_HIP_CHECK(hipMemcpy(d, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
    int r = launch_flash_attn(Q, ld, K, V, ld, O, ld, d, (int)tiles.size(), scale * 1.4426950408889634f, st);
    hipStreamSynchronize(st);     // test entry point only: the tile table is freed right away
    hipFree(d);
    return r;
}

int vlsat_k_flash_attn_bf16(const float* Q, const float* K, const float* V, float* O, int32_t ld, const int64_t* tok_ptr,
                            int32_t n_scenes, int32_t n_heads, float scale, int32_t terms, int32_t use_tr, void* stream) {
    if (!tok_ptr || n_scenes <= 0) return fail(VLSAT_EINVAL, "flash_attn: bad scene table");
    std::vector<int4> tiles;
    for (int s = 0; s < n_scenes; ++s) {
        const int64_t T = tok_ptr[s + 1] - tok_ptr[s];
        for (int hh = 0; hh < n_heads; ++hh)
            for (int64_t q0 = 0; q0 < T; q0 += FLASH_BQ) tiles.push_back(make_int4((int)tok_ptr[s], (int)T, (int)q0, hh));
    }
    if (tiles.empty()) return 0;
    int4* d = nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d), tiles.size() * sizeof(int4)));
    VLSAT_HIP_CHECK(hipMemcpy(d, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice));
    int r = launch_flash_attn_bf16(Q, ld, K, V, ld, O, ld, d, (int)tiles.size(), scale * 1.4426950408889634f, terms, use_tr != 0,
                                   use_tr == 2 ? 1 : use_tr == 3 ? 2 : 0, st);
    hipStreamSynchronize(st);     // test entry point only: the tile table is freed right away
    hipFree(d);
    return r;
}

int vlsat_prepare_objects(const float* scene_points, const int32_t* choice, int32_t n_obj, int32_t n_points,
                          float* obj_points, float* descriptor, void* stream) {
    if (!scene_points || !choice || !obj_points || !descriptor) return fail(VLSAT_EINVAL, "prepare_objects: null argument");
    return launch_prepare_objects(scene_points, choice, n_obj, n_points, obj_points, descriptor, static_cast<hipStream_t>(stream));
}

int vlsat_fc_edges(const int32_t* node_ptr, const int64_t* edge_ptr, int32_t n_scenes, int64_t n_nodes, int64_t n_edges,
                   int64_t* edges, int64_t* batch_ids, void* stream) {
    if (!node_ptr || !edge_ptr || !batch_ids || (n_edges > 0 && !edges) || n_scenes <= 0)
        return fail(VLSAT_EINVAL, "fc_edges: bad argument");
    return launch_fc_edges(node_ptr, edge_ptr, n_scenes, n_nodes, n_edges, edges, batch_ids, static_cast<hipStream_t>(stream));
}

int vlsat_k_softmax_rows(const float* x, int32_t ld, int32_t rows, int32_t cols, float* out, void* stream) {
    if (!x || !out) return fail(VLSAT_EINVAL, "softmax_rows: null argument");
    return launch_softmax_rows(x, ld, rows, cols, out, 0, static_cast<hipStream_t>(stream));
}

int vlsat_eval_ranks(const float* obj_logits, const float* obj_probs, const float* rel_probs, const int64_t* gt_class,
                     const int64_t* gt_rel, const int64_t* edges, int32_t n_nodes, int32_t n_edges, int32_t n_obj_class,
                     int32_t n_rel_class, int32_t topk_obj, int32_t topk_rel, int32_t topk_triplet, float threshold,
                     int32_t* obj_rank, int32_t* rel_rank, int32_t* tri_rank, int32_t* cnt, void* stream) {
    if (!obj_logits || !obj_probs || !gt_class || !obj_rank) return fail(VLSAT_EINVAL, "eval_ranks: null argument");
    if (n_edges > 0 && (!rel_probs || !gt_rel || !edges || !rel_rank || !tri_rank || !cnt))
        return fail(VLSAT_EINVAL, "eval_ranks: null edge argument");
    return launch_eval_ranks(obj_logits, obj_probs, rel_probs, gt_class, gt_rel, edges, n_nodes, n_edges, n_obj_class,
                             n_rel_class, topk_obj, topk_rel, topk_triplet, threshold, obj_rank, rel_rank, tri_rank, cnt,
                             static_cast<hipStream_t>(stream));
}

int vlsat_scene_checksums(const float* obj3d, const float* obj2d, int64_t n_nodes, int32_t n_obj_class, const float* rel3d,
                          const float* rel2d, int64_t n_edges, int32_t n_rel_class, int32_t n_scenes, double* out9, double* scratch,
                          void* stream) {
    if (!obj3d || !obj2d || !out9 || !scratch || n_nodes < 0 || n_edges < 0) return fail(VLSAT_EINVAL, "scene_checksums: null argument");
    if (n_edges > 0 && (!rel3d || !rel2d)) return fail(VLSAT_EINVAL, "scene_checksums: null relation outputs");
    return launch_scene_checksums(obj3d, obj2d, (long)n_nodes, n_obj_class, rel3d, rel2d, (long)n_edges, n_rel_class, n_scenes, out9,
                                  scratch, static_cast<hipStream_t>(stream));
}

int vlsat_k_layernorm(float* x, int32_t ld, int32_t rows, int32_t dim, const float* gamma, const float* beta,
                      int32_t relu, void* stream) {
    return launch_layernorm(x, ld, rows, dim, gamma, beta, relu, static_cast<hipStream_t>(stream));
}

}  // extern "C"
