// Kernel-level C entry points (unit tests and tools drive single kernels through these) and debug hooks.
#include <algorithm>
#include <climits>
#include <cstdint>
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

// Switches of one handle (none changes results beyond fp32 summation order / the mode's rounding; defaults are the measured-best
// settings).  The RELEASE library knows the ones a deployment or a test has a use for:
//   "dual_stream" 0|1|2 a second (and third) stream for the 2D chains: 1 small plans only, 2 every plan (plans created afterwards)
//   "sched"       -1|0|1  two-stream plans: dependency-exact three-lane schedule (1), the fork / join schedule of round 4 (0), or by
//                       mode (-1, default: exact in the bf16 modes, fork / join in exact fp32)
//   "flash_split" 0|1   split-key edge attention for plans that cannot fill the chip (plans created afterwards)
//   "prof_dual"   0|1   per-class profiling keeps the multi-stream execution (1, default) or serialises on the launch stream
//   "pair_twins" 0|1 / "pair_max_edges" n   paired schedule of one-scene plans (engine_forward.hip): twin stages as launches of two problems
//   "gather_f16" -1|0|1    [P_i | P_j] of the node-side projection as fp16 half rows (-1: on in the single-rounding modes)
//   "outproj_f16" -1|0|1   out-projection of the single-rounded edge attention -> LayerNorm as fp16 half rows instead of fp32 (-1: on)
//   "gemm_k_rot" -1|0..7   K-tile rotation per column tile of the 8-phase GEMM (-1: 1 for half-row bf16 launches, else 0)
//   "gemm_p8" / "gemm_dma" / "gemm_splitk" 0|1   GEMM kernel selection: 256 x 256 8-phase kernel for large launches, LDS-direct staging of
//                       fp32 operands, split-K kernel for small launches (0: the older kernels; parity-tested both ways)
//   "split_fmt" / "flash_bf16" / "flash_tr" / "pointnet_bf16" / "gate_bf16" / "ln_resid" 0|1   bf16 modes: edge tensors between matrix
//                       kernels as bf16 hi/lo pairs or half rows, attention / object encoder / gate on the bf16 matrix cores, V operand by
//                       LDS transpose read, attention residual added by the LayerNorm kernel (0: the fp32 forms; parity-tested both ways)
//   "flash_pv_terms" 3|2   split-bf16 edge attention: MFMAs per P.V product
//   "flash_bq_big" 0|1   half-row edge attention on plans whose scenes all have >= 4096 edges: 256 queries per block (1, default) or 128
//   "gate_fuse_agg" 0|1|2  max aggregation inside the gate kernel: never / bf16 modes (default) / fp32 too
//   "gate_row_map" 0|1, "gate_heads_mfma" 0|1|2   gate kernel variants the tests compare bit for bit
// An EXPERIMENTS build (build.py --experiments, -DVLSAT_EXPERIMENTS) also accepts the lab switches -- "gate_grid" n, "gate_heads_bf16",
// "flash_heads_bf16", "node_attn_split" n, "half_fmt", "flash_dma" 0|1|3|4, "flash_ablate" bits (GARBAGE results: timing only) -- and
// honours the ablation bits of GemmArgs / FlashSplit; the release library answers them with an error.
int vlsat_debug_option(vlsat_handle h, const char* name, int32_t value) {
    if (!h || !name) return fail(VLSAT_EINVAL, "vlsat_debug_option: null argument");
    const std::string k(name);
    ++h->config_epoch;
    if (k == "dual_stream") h->dual_stream = value < 0 ? 0 : value;
    else if (k == "sched") h->sched = value < 0 ? -1 : value != 0;
    else if (k == "flash_split") h->fa_split = value != 0;
    else if (k == "gemm_dma") h->gemm_no_dma = value == 0;
    else if (k == "gemm_p8") h->gemm_no_p8 = value == 0;
    else if (k == "pair_twins") h->pair_twins = value != 0;
    else if (k == "pair_max_edges") h->pair_max_edges = value < 0 ? 0 : value;
    else if (k == "gather_f16") h->gather_f16 = value < 0 ? -1 : value != 0;
    else if (k == "outproj_f16") h->outproj_f16 = value < 0 ? -1 : value != 0;
    else if (k == "gemm_k_rot") h->gemm_k_rot = value < 0 ? -1 : value > 7 ? 7 : value;
    else if (k == "gate_row_map") h->gate_row_map = value != 0;
    else if (k == "prof_dual") h->prof_dual = value != 0;
    else if (k == "gate_heads_mfma") h->gate_heads_mfma = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (k == "gemm_splitk") h->gemm_splitk = value != 0;
    else if (k == "gemm_p8_part_min") h->gemm_p8_part_min = value < 0 ? 0 : value;
    else if (k == "gemm_splitk_max_tiles") h->gemm_splitk_max_tiles = value < 0 ? 0 : value;
    else if (k == "split_fmt") h->split_fmt = value != 0;
    else if (k == "ln_resid") h->ln_resid = value != 0;
    else if (k == "gate_fuse_agg") h->gate_fuse_agg = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (k == "flash_pv_terms") h->flash_pv_terms = value == 2 ? 2 : 3;
    else if (k == "flash_bf16") h->flash_bf16 = value != 0;
    else if (k == "pointnet_bf16") h->pointnet_bf16 = value != 0;
    else if (k == "gate_bf16") h->gate_bf16 = value != 0;
    else if (k == "flash_tr") h->flash_tr = value != 0;
    else if (k == "flash_bq_big") h->flash_bq_big = value != 0;
    else if (k == "flash_qg") h->flash_qg = value < 0 || value > 2 ? 0 : value;
    else if (k == "flash_bq_big_min") h->flash_bq_big_min = value < 256 ? 256 : value;
#ifdef VLSAT_EXPERIMENTS
    else if (k == "gate_grid") h->gate_grid = value > 0 ? value : 0;
    else if (k == "gate_heads_bf16") h->gate_heads_bf16 = value != 0;
    else if (k == "flash_heads_bf16") h->flash_heads_bf16 = value != 0;
    else if (k == "node_attn_split") h->node_attn_split = value;
    else if (k == "half_fmt") h->half_fmt = value != 0;
    else if (k == "flash_dma") h->flash_dma = (value == 3 || value == 4) ? value : value != 0;
    else if (k == "flash_ablate") h->flash_ablate = value;
#else
    else if (k == "gate_grid" || k == "gate_heads_bf16" || k == "flash_heads_bf16" || k == "node_attn_split" || k == "half_fmt" ||
             k == "flash_dma" || k == "flash_ablate")
        return fail(VLSAT_EINVAL, "vlsat_debug_option: " + k + " is a lab switch of the experiments build (build.py --experiments)");
#endif
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
    if (kExperiments) {
        a.force_tile = (relu_a >> 4) & 7;       // (bits 4..6: tile of gemm_f32_kernel, GemmArgs::force_tile -- benchmarking; experiments build)
        if (a.force_tile) { a.no_p8 = 1; a.no_ring = 1; }
    }
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
    // (bit 3: g0 / g1 are fp16 half rows; bits 25..28: that many times 256 leading columns of C are written as fp16 half rows -- GemmArgs::g_f16,
    //  c_f16_cols; bit 4: no ring kernel -- benchmarking)
    a.g_f16 = (fmt >> 3) & 1;
    a.c_f16_cols = ((fmt >> 25) & 15) * 256;
    a.no_ring = (fmt >> 4) & 1;
    a.no_p8 = (fmt >> 12) & 1;                 // (bit 12: half-row launches skip the 256 x 256 8-phase kernel)
    a.k_rot = (fmt >> 22) & 7;                 // (bits 22..24: K-tile rotation per column tile of the 8-phase kernel -- A/B)
    if (kExperiments) {                        // lab bits (tools/gemm_bench.py, p8_check.py, gemm_tile_sweep.py): the experiments build only
        a.ring_wide = (fmt >> 7) & 1;              // (bit 7: ring kernel with 128 x 256 tiles where N allows)
        a.ablate = ((fmt >> 8) & 3) | (((fmt >> 13) & 63) << 2);   // (bits 8, 9, 13..18: timing experiments, see GemmArgs::ablate)
        a.ring_bk32 = (fmt >> 10) & 1;             // (bit 10: half-row ring kernel with 32-wide k slices)
        a.ring_nodb = (fmt >> 11) & 1;             // (bit 11: ... without the double-buffered fragment sets)
        a.force_tile = (fmt >> 19) & 7;            // (bits 19..21: tile of gemm_f32_kernel, GemmArgs::force_tile -- benchmarking)
        if (a.force_tile) { a.no_p8 = 1; a.no_ring = 1; }
    } else if (fmt & ((1 << 7) | (3 << 8) | (3 << 10) | (0x1ff << 13))) {
        return fail(VLSAT_EINVAL, "gemm_planes: timing / tile experiment bits need the experiments build (build.py --experiments)");
    }
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
    FlashSplit sp;                              // (no split keys; rows = the tensors' rows: lets the half-row variant take its LDS-direct staging)
    sp.rows = (int)tok_ptr[n_scenes];
    int r = launch_flash_attn_bf16(Q, ld, K, V, ld, O, ld, d, (int)tiles.size(), scale * 1.4426950408889634f, terms, use_tr != 0,
                                   use_tr == 2 ? 1 : use_tr == 3 ? 2 : 0, st, &sp);
    hipStreamSynchronize(st);     // test entry point only: the tile table is freed right away
    hipFree(d);
    return r;
}

int vlsat_prepare_objects(const float* scene_points, const int32_t* choice, int32_t n_obj, int32_t n_points,
                          float* obj_points, float* descriptor, void* stream) {
    if (!scene_points || !choice || !obj_points || !descriptor) return fail(VLSAT_EINVAL, "prepare_objects: null argument");
    return launch_prepare_objects(scene_points, choice, n_obj, n_points, obj_points, descriptor, static_cast<hipStream_t>(stream));
}

// Per-object point selection on the device (the step in front of vlsat_prepare_objects; reference dataset_3dssg.py:279-289): see
// include/vlsat.h.  vlsat_sample_objects_scratch gives the int32 count of `scratch`.
size_t vlsat_sample_objects_scratch(int64_t n_points, int32_t n_obj) { return sample_objects_scratch_ints(n_points, n_obj); }
int vlsat_sample_objects(const int32_t* instances, int64_t n_points, const int32_t* instance_ids, int32_t n_obj, int32_t n_sample,
                         uint64_t seed, int32_t* id_map, int32_t map_size, int32_t* scratch, int32_t* choice, int32_t* counts, void* stream) {
    if (!instances || !instance_ids || !id_map || !scratch || !choice || !counts) return fail(VLSAT_EINVAL, "sample_objects: null argument");
    return launch_sample_objects(instances, n_points, instance_ids, n_obj, n_sample, seed, id_map, map_size, scratch, choice, counts,
                                 static_cast<hipStream_t>(stream));
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
                     int32_t* obj_rank, int32_t* rel_rank, int32_t* tri_rank, int32_t* cnt, float* scratch, void* stream) {
    if (!obj_logits || !obj_probs || !gt_class || !obj_rank) return fail(VLSAT_EINVAL, "eval_ranks: null argument");
    if (n_edges > 0 && (!rel_probs || !gt_rel || !edges || !rel_rank || !tri_rank || !cnt || !scratch))
        return fail(VLSAT_EINVAL, "eval_ranks: null edge argument");
    return launch_eval_ranks(obj_logits, obj_probs, rel_probs, gt_class, gt_rel, edges, n_nodes, n_edges, n_obj_class,
                             n_rel_class, topk_obj, topk_rel, topk_triplet, threshold, obj_rank, rel_rank, tri_rank, cnt, scratch,
                             static_cast<hipStream_t>(stream));
}

int64_t vlsat_eval_ranks_scratch_floats(int32_t n_nodes, int32_t n_obj_class, int32_t topk_triplet) {
    if (n_nodes < 0 || n_obj_class < 0 || topk_triplet < 0) return 0;
    return (int64_t)n_nodes * eval_ranks_sorted_k(n_obj_class, topk_triplet);
}

int vlsat_eval_counts(const int32_t* obj_rank_3d, const int32_t* obj_rank_2d, const int32_t* rel_rank_3d, const int32_t* rel_rank_2d,
                      const int32_t* tri_rank_3d, const int32_t* tri_rank_2d, const int32_t* cnt, const int64_t* gt_class,
                      const int64_t* gt_rel, const int64_t* edges, int32_t n_nodes, int32_t n_edges, int32_t n_rel_class,
                      int32_t n_scenes, uint64_t* counts, void* stream) {
    if (!counts || n_nodes < 0 || n_edges < 0) return fail(VLSAT_EINVAL, "eval_counts: bad argument");
    if (n_nodes > 0 && (!obj_rank_3d || !obj_rank_2d || !gt_class)) return fail(VLSAT_EINVAL, "eval_counts: null node argument");
    if (n_edges > 0 && (!rel_rank_3d || !rel_rank_2d || !tri_rank_3d || !tri_rank_2d || !cnt || !gt_rel || !edges || !gt_class || !obj_rank_3d))
        return fail(VLSAT_EINVAL, "eval_counts: null edge argument");
    return launch_eval_counts(obj_rank_3d, obj_rank_2d, rel_rank_3d, rel_rank_2d, tri_rank_3d, tri_rank_2d, cnt, gt_class, gt_rel, edges,
                              n_nodes, n_edges, n_rel_class, n_scenes, reinterpret_cast<unsigned long long*>(counts),
                              static_cast<hipStream_t>(stream));
}

// One scene (or batch) of an evaluation loop in ONE call: forward + softmax of the object logits + both ranking passes + the
// additive counts (vlsat_forward, vlsat_k_softmax_rows, vlsat_eval_ranks x 2, vlsat_eval_counts), all enqueued on `stream`,
// intermediates in the plan's own scratch -- what Mmgnet.process_val (reference SGFN_MMG/model.py:458-480) computes per scene,
// reduced to what MMGNet.validation keeps of it.  The host side of a one-scene-per-call loop is then one library call per scene
// instead of six plus a dozen tensor allocations.  gt_rel: the multi-hot [E, R] int64 target; edges_e2: [E, 2] int64 (from, to) as
// the loader yields it, in the plan's edge order; counts: device uint64 [1 + R + 2 (11 + 6 R)], zeroed once by the caller.
// Top-k bounds and threshold are process_val's constants (topk 11 / 6 / 101, 0.5: reference :463-472).
// MODEL.multi_rel_outputs only (the single-label variant ranks exp(log-softmax) in a second pass: use the separate entry points).
int vlsat_process_val_counts(vlsat_handle h, vlsat_plan p, const float* obj_points, const float* obj_2d_feats, const float* descriptor,
                             const int64_t* gt_class, const int64_t* gt_rel, const int64_t* edges_e2, int32_t n_scenes,
                             uint64_t* counts, void* stream) {
    if (!h || !p || !gt_class || !counts) return fail(VLSAT_EINVAL, "vlsat_process_val_counts: null argument");
    if (p->h != h) return fail(VLSAT_EINVAL, "plan belongs to a different handle");
    if (!h->d.multi_rel_outputs) return fail(VLSAT_EINVAL, "vlsat_process_val_counts: built for MODEL.multi_rel_outputs (use vlsat_forward + vlsat_eval_ranks + vlsat_eval_counts)");
    const int N = (int)p->N, E = (int)p->E, C = h->d.n_obj_class, R = h->d.n_rel_class;
    if (E > 0 && (!gt_rel || !edges_e2)) return fail(VLSAT_EINVAL, "vlsat_process_val_counts: null edge argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* f = p->ev_f;
    float *obj3 = f, *obj2 = f + (size_t)N * C, *prob3 = f + (size_t)2 * N * C, *prob2 = f + (size_t)3 * N * C;
    float *rel3 = f + (size_t)4 * N * C, *rel2 = rel3 + (size_t)std::max(E, 1) * R;
    float* sorted = rel2 + (size_t)std::max(E, 1) * R;          // [N, min(C, 101)]: the ranking's per-node sorted probabilities
    int32_t* i = p->ev_i;
    int32_t *or3 = i, *or2 = i + N, *rr3 = i + 2 * (size_t)N, *rr2 = rr3 + (size_t)std::max(E, 1) * R, *tr3 = rr2 + (size_t)std::max(E, 1) * R,
            *tr2 = tr3 + (size_t)std::max(E, 1) * R, *cn3 = tr2 + (size_t)std::max(E, 1) * R, *cn2 = cn3 + std::max(E, 1);
    RUN(vlsat_forward(h, p, obj_points, obj_2d_feats, descriptor, obj3, obj2, rel3, rel2, stream));
    for (int br = 0; br < 2; ++br) {
        const float* lg = br ? obj2 : obj3;
        float* pr = br ? prob2 : prob3;
        RUN(launch_softmax_rows(lg, C, N, C, pr, 0, s));
        RUN(launch_eval_ranks(lg, pr, br ? rel2 : rel3, gt_class, gt_rel, edges_e2, N, E, C, R, 11, 6, 101, 0.5f, br ? or2 : or3, br ? rr2 : rr3,
                              br ? tr2 : tr3, br ? cn2 : cn3, sorted, s));
    }
    RUN(launch_eval_counts(or3, or2, rr3, rr2, tr3, tr2, cn3, gt_class, gt_rel, edges_e2, N, E, R, n_scenes,
                           reinterpret_cast<unsigned long long*>(counts), s));
    VLSAT_HIP_CHECK(hipEventRecord(p->last_use, s));       // (the scratch is the plan's: its next owner orders behind the counting)
    return 0;
}

int vlsat_scene_checksums(const float* obj3d, const float* obj2d, int64_t n_nodes, int32_t n_obj_class, const float* rel3d,
                          const float* rel2d, int64_t n_edges, int32_t n_rel_class, int32_t n_scenes, double* out9, double* scratch,
                          void* stream) {
    if (!obj3d || !obj2d || !out9 || !scratch || n_nodes < 0 || n_edges < 0) return fail(VLSAT_EINVAL, "scene_checksums: null argument");
    if (n_edges > 0 && (!rel3d || !rel2d)) return fail(VLSAT_EINVAL, "scene_checksums: null relation outputs");
    return launch_scene_checksums(obj3d, obj2d, (long)n_nodes, n_obj_class, rel3d, rel2d, (long)n_edges, n_rel_class, n_scenes, out9,
                                  scratch, static_cast<hipStream_t>(stream));
}

// -------------------------------------------------------------------------------------------
// Kernel-level entry points of the non-GEMM kernels of the GCN block and the node attention (SURVEY 8b's per-kernel list):
// test entry points like vlsat_k_flash_attn -- small index tables are built on the host, uploaded, and the call
// synchronises before it frees them.
namespace {
struct DevTable {                       // a host table copied to the device for the duration of one call
    void* d = nullptr;
    ~DevTable() { if (d) hipFree(d); }
    int put(const void* host, size_t bytes) {
        if (!bytes) return 0;
        VLSAT_HIP_CHECK(hipMalloc(&d, bytes));
        VLSAT_HIP_CHECK(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
        return 0;
    }
};
// node offsets of the scenes (int64 host, like batch_ids run lengths) -> int32 scene_ptr, int64 bias_ptr, largest scene
static int scene_tables(const int64_t* node_ptr, int n_scenes, int n_heads, std::vector<int32_t>& sp, std::vector<int64_t>& bp, int* max_n) {
    if (!node_ptr || n_scenes <= 0) return fail(VLSAT_EINVAL, "bad scene table");
    sp.resize(n_scenes + 1);
    bp.resize(n_scenes);
    int64_t off = 0;
    *max_n = 0;
    for (int s = 0; s <= n_scenes; ++s) {
        if (node_ptr[s] < 0 || node_ptr[s] > INT32_MAX || (s && node_ptr[s] < node_ptr[s - 1])) return fail(VLSAT_EINVAL, "scene table must be ascending int32 offsets");
        sp[s] = (int32_t)node_ptr[s];
        if (s < n_scenes) {
            const int64_t n = node_ptr[s + 1] - node_ptr[s];
            bp[s] = off;
            off += (int64_t)n_heads * n * n;
            if (n > *max_n) *max_n = (int)n;
        }
    }
    return 0;
}
}  // namespace

int vlsat_k_edge_gate(const float* kproj, const float* node, int32_t ld_node, int32_t gq_off, int32_t v_off, const int32_t* src,
                      const int32_t* dst, const float* w0k, const float* w3, const float* b3, float* gated, float* prob,
                      int32_t n_edges, int32_t n_heads, int32_t dk, int32_t dox, int32_t use_edge, int32_t variant, void* stream) {
    if (!node || !src || !dst || !w3 || !b3 || !gated || (use_edge && (!kproj || !w0k))) return fail(VLSAT_EINVAL, "edge_gate: null argument");
    if (n_edges < 0 || n_heads <= 0 || dk <= 0 || dox <= 0) return fail(VLSAT_EINVAL, "edge_gate: bad geometry");
    hipStream_t s = static_cast<hipStream_t>(stream);
    GateArgs g{};
    g.kproj = kproj; g.node = node; g.ld_node = ld_node; g.gq_off = gq_off; g.v_off = v_off; g.src = src; g.dst = dst;
    g.w0k = w0k; g.w3 = w3; g.b3 = b3; g.gated = gated; g.prob = prob; g.n_edges = n_edges; g.use_edge = use_edge != 0;
    const bool def = n_heads == 8 && dk == 64 && dox == 32;
    switch (variant) {
        case 0:                                                  // what the fp32 forward launches for this geometry
            if (def) return launch_edge_gate(g, s);
            {
                const int r = launch_edge_gate_heads(g, n_heads, dk, dox, s);
                if (r <= 0) return r;
            }
            return launch_edge_gate_generic(g, n_heads, dk, dox, s);
        case 1: return launch_edge_gate_generic(g, n_heads, dk, dox, s);
        case 2: {
            const int r = launch_edge_gate_heads(g, n_heads, dk, dox, s);
            return r > 0 ? fail(VLSAT_EINVAL, "edge_gate: head geometry not built on the MFMA template") : r;
        }
        case 3: case 4: {
            const int terms = variant == 3 ? 3 : 1;
            if (def) return launch_edge_gate_bf16(g, terms, 0, s);
            if (!edge_gate_bf16_heads_supports(dk, dox, terms)) return fail(VLSAT_EINVAL, "edge_gate: head geometry / terms not built on the bf16 template");
            const int r = launch_edge_gate_bf16_heads(g, n_heads, dk, dox, terms, 0, s);
            return r > 0 ? fail(VLSAT_EINVAL, "edge_gate: head geometry not built on the bf16 template") : r;
        }
        default: return fail(VLSAT_EINVAL, "edge_gate: variant 0..4");
    }
}

int vlsat_k_aggregate(const float* gated, int32_t n_ch, const int64_t* index_host, int64_t n_edges, int32_t n_nodes, int32_t aggr,
                      float* out, int32_t ldo, int32_t col0, void* stream) {
    if (!out || n_nodes < 0 || n_edges < 0 || n_ch <= 0 || (n_edges > 0 && (!gated || !index_host))) return fail(VLSAT_EINVAL, "aggregate: bad argument");
    if (aggr < 0 || aggr > 2) return fail(VLSAT_EINVAL, "aggregate: aggr 0 max | 1 add | 2 mean");
    if (n_edges > INT32_MAX) return fail(VLSAT_EINVAL, "aggregate: too many rows");
    std::vector<int32_t> rowptr(n_nodes + 1, 0), order((size_t)n_edges);
    for (int64_t e = 0; e < n_edges; ++e) {
        if (index_host[e] < 0 || index_host[e] >= n_nodes) return fail(VLSAT_EINVAL, "aggregate: index out of range");
        ++rowptr[index_host[e] + 1];
    }
    for (int n = 0; n < n_nodes; ++n) rowptr[n + 1] += rowptr[n];
    std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
    for (int64_t e = 0; e < n_edges; ++e) order[fill[index_host[e]]++] = (int32_t)e;      // stable: rows of a segment in input order
    DevTable dr, dord;
    RUN(dr.put(rowptr.data(), rowptr.size() * sizeof(int32_t)));
    RUN(dord.put(order.data(), order.size() * sizeof(int32_t)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int r = launch_aggregate(gated, n_ch, static_cast<const int32_t*>(dr.d), static_cast<const int32_t*>(dord.d), n_nodes, aggr, out, ldo, col0, s);
    hipStreamSynchronize(s);
    return r;
}

int vlsat_k_node_attn(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv, float* O, int32_t ldo,
                      const float* bias, const int64_t* node_ptr_host, int32_t n_scenes, int32_t n_heads, float scale,
                      int32_t lanes_per_query, void* stream) {
    if (!Q || !K || !V || !O) return fail(VLSAT_EINVAL, "node_attn: null argument");
    if (n_heads <= 0 || 512 % n_heads) return fail(VLSAT_EINVAL, "node_attn: n_heads must divide 512");
    if (lanes_per_query != 0 && lanes_per_query != 1 && lanes_per_query != 16) return fail(VLSAT_EINVAL, "node_attn: lanes_per_query 0 (auto) | 1 | 16");
    std::vector<int32_t> sp;
    std::vector<int64_t> bp;
    int max_n = 0;
    RUN(scene_tables(node_ptr_host, n_scenes, n_heads, sp, bp, &max_n));
    DevTable dsp, dbp;
    RUN(dsp.put(sp.data(), sp.size() * sizeof(int32_t)));
    RUN(dbp.put(bp.data(), bp.size() * sizeof(int64_t)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int split_below = lanes_per_query == 16 ? INT32_MAX : lanes_per_query == 1 ? 0 : 1024;
    const int r = launch_node_attn(Q, ldq, K, ldk, V, ldv, O, ldo, bias, static_cast<const int32_t*>(dsp.d), static_cast<const int64_t*>(dbp.d),
                                   n_scenes, max_n, n_heads, 512 / n_heads, scale, s, split_below);
    hipStreamSynchronize(s);
    return r;
}

int vlsat_k_dist_bias(const float* desc, int32_t ld_desc, const int64_t* node_ptr_host, int32_t n_scenes, int32_t n_heads,
                      const float* w0, const float* b0, const float* ln2_w, const float* ln2_b, const float* w3, const float* b3,
                      const float* ln5_w, const float* ln5_b, const float* w6, const float* b6, float* bias, void* stream) {
    if (!desc || !w0 || !b0 || !ln2_w || !ln2_b || !w3 || !b3 || !ln5_w || !ln5_b || !w6 || !b6 || !bias) return fail(VLSAT_EINVAL, "dist_bias: null argument");
    if (n_heads <= 0 || ld_desc < 3) return fail(VLSAT_EINVAL, "dist_bias: bad geometry");
    std::vector<int32_t> sp;
    std::vector<int64_t> bp;
    int max_n = 0;
    RUN(scene_tables(node_ptr_host, n_scenes, n_heads, sp, bp, &max_n));
    DevTable dsp, dbp;
    RUN(dsp.put(sp.data(), sp.size() * sizeof(int32_t)));
    RUN(dbp.put(bp.data(), bp.size() * sizeof(int64_t)));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const DistBiasW w{w0, b0, ln2_w, ln2_b, w3, b3, ln5_w, ln5_b, w6, b6};
    const int r = launch_dist_bias(desc, ld_desc, static_cast<const int32_t*>(dsp.d), static_cast<const int64_t*>(dbp.d), n_scenes, max_n, n_heads, w, bias, s);
    hipStreamSynchronize(s);
    return r;
}

int vlsat_k_layernorm(float* x, int32_t ld, int32_t rows, int32_t dim, const float* gamma, const float* beta,
                      int32_t relu, void* stream) {
    return launch_layernorm(x, ld, rows, dim, gamma, beta, relu, static_cast<hipStream_t>(stream));
}

}  // extern "C"
