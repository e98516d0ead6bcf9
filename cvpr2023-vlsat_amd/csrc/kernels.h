// Host-side launchers of the hand-written gfx950 kernels (internal C++ interface between the
// engine and the .hip files; the public surface is include/vlsat.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlsat {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

// C[M,N] = act(rowscale[m]*(reluA?(A) . W^T) + bias[n] + resid_scale*resid[m,n] + g0[gi0[m],n] + g1[gi1[m],n])
struct GemmArgs {
    const float* A = nullptr; int lda = 0;      // [M,K]
    const float* W = nullptr; int ldw = 0;      // [N,K]  (nn.Linear layout)
    float* C = nullptr;       int ldc = 0;      // [M,N]
    int M = 0, N = 0, K = 0;
    const float* bias = nullptr;                // [N]
    const float* rowscale = nullptr;            // [M]
    const float* resid = nullptr; int ldr = 0; float resid_scale = 1.f;
    const float* g0 = nullptr; const int32_t* gi0 = nullptr; int ldg0 = 0;   // gathered row add
    const float* g1 = nullptr; const int32_t* gi1 = nullptr; int ldg1 = 0;
    int relu_a = 0;                             // apply ReLU to A while staging
    int act = ACT_NONE;
    // split-bf16 path: prec 0 = exact fp32 MFMA, 1 = bf16, 3 = bf16x3; weights pre-split [N,K] bf16 (ldw shared)
    int prec = 0;
    const uint16_t* Whi = nullptr;
    const uint16_t* Wlo = nullptr;
    long long* clock_probe = nullptr;           // optional [grid][4] DVFS probe buffer (vlsat_debug_gemm_clock_probe)
    // storage format of A / the residual / C: 0 fp32, 1 split-pair words (common.h pack_split; split-bf16 mode),
    // 2 half rows (bf16 values at byte 2 * column of an fp32-pitched row; single-rounding modes)
    int a_split = 0, r_split = 0, c_split = 0;
    float c_scale = 1.f;                        // final multiplier of C (after bias / activation)
    int ablate = 0;                             // timing experiments on the ring kernel: bit 0 no operand loads after the first slices, bit 1 no MFMAs, bit 2 (8-phase kernel) no fragment reads (results are garbage)
    int ring_nodb = 0;                          // experiment: half-row ring kernel without the double-buffered fragment sets
    int ring_bk32 = 0;                          // experiment: half-row ring kernel with 32-wide k slices (default 64 where K allows)
    int ring_wide = 0;                          // experiment: bf16 ring kernel with 128 x 256 tiles where N allows (measured equal)
    int no_ring = 0;                            // debug: keep large bf16 launches on the two-stage 128 x 128 kernel
    int no_p8 = 0;                              // debug: large launches skip the 256 x 256 8-phase kernel (gemm_bf16_p8.hip)
    int p8_part_min = 0;                        // 8-phase kernel: tiles from which a remainder rides along as balanced rounds / a partial round (0: the built-in bound, 32 in bf16, 5/8 of a round otherwise)
    int sk_max_tiles = 0;                       // split-K kernel only for launches of at most this many 64 x 64 tiles (0: half the resident slots, the rule of rounds 2-5)
    int k_rot = 0;                              // A-B: the column tiles of a row panel walk their K-tiles rotated by tn * k_rot (8-phase kernel: siblings re-read the A panel out of step)
    // fp16 additive tables (round 6; the single-rounding modes): the first c_f16_cols columns of C (a multiple of the block tile's width) are
    // stored as fp16 HALF ROWS (element n at byte 2 n of the fp32-pitched row, values clamped to +-65504) -- what the node-side projection
    // writes for [P_i | P_j]; g_f16: g0 / g1 are such half rows (launches without a residual).  Halves the bytes nn_edge.0 gathers per edge.
    int c_f16_cols = 0, g_f16 = 0;
    // fp16 half-row OPERANDS (precision mode "fp16_mixed"): A (a_split == 2) holds fp16 instead of bf16, Whi is an fp16 plane, the products run on
    // v_mfma_f32_32x32x16_f16 (same rate as bf16 on CDNA4, 2^-12 instead of 2^-9 per operand); half-row outputs then go through c_f16_cols == N
    int half_f16 = 0;
    int force_tile = 0;                         // experiment (tools/gemm_tile_sweep.py): 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64 tiles of gemm_f32_kernel, whatever the heuristic says
    int prefetch = -1;                          // bf16 LDS-direct pipe: slices of look-ahead of the A-panel prefetch (0 off, -1 default)
    int no_dma = 0;                             // debug: VGPR-staged fp32 operands instead of LDS-direct (vlsat_debug_option "gemm_dma")
    long* launches = nullptr;                   // optional host counter, +1 per kernel launched (profiling)
    // split-K path of small launches (gemm_splitk.hip): partial-sum workspace + per-tile arrival counters (zero between
    // launches), owned by the caller and private to the stream the launch goes to; null = never split
    float* sk_ws = nullptr; size_t sk_ws_floats = 0;
    unsigned* sk_counters = nullptr; size_t sk_n_counters = 0;
};
int launch_gemm(const GemmArgs& a, hipStream_t s);
// small launches: k range cut over several CUs, deterministic in-kernel reduction; 1 = not applicable, 0 = launched
int launch_gemm_splitk(const GemmArgs& a, int slots, hipStream_t s, const GemmArgs* twin = nullptr);
// two problems of the same shape and flags in ONE launch (round 6: the 3D / 2D twins of a one-scene forward): 0 = launched,
// 1 = not pairable (different shapes / flags, or a launch the single-round small-tile kernels would not take): the caller
// launches them one after the other; results are bit-identical either way
int launch_gemm_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t s);
constexpr size_t SPLITK_WS_FLOATS = (size_t)768 * 4096;    // room for 768 partial 64 x 64 tiles (12 MB)
constexpr size_t SPLITK_COUNTERS = 512;
// bf16 modes, full rounds of large-M launches: 3-stage LDS ring, 256 x 128 tiles, one 8-wave block per CU
// (gemm_bf16_ring.hip); returns 1 if the operand combination is not built
int launch_gemm_ring(const GemmArgs& a, int rbn, int n_tiles, int grid, hipStream_t s);   // rbn: tile width 128 | 256
// full rounds of large-M launches, exact fp32 or single-rounding bf16 with half-row A: 256 x 256 tiles, 8-phase pipeline,
// one 8-wave block per CU (gemm_bf16_p8.hip); needs M % 256 == 0, N % 256 == 0, K % 128 == 0; returns 1 if the combination is not built
int launch_gemm_p8(const GemmArgs& a, int n_tiles, int grid, hipStream_t s);
double gemm_flops(const GemmArgs& a);
void gemm_set_clock_probe(long long* buf);

// ---- PointNet object encoder (fused conv1..conv3 + ReLU + max over points) ----
int launch_pointnet(const float* pts, int n_obj, int n_points, int cin, const float* w1, const float* b1,
                    const float* w2, const float* b2, const float* w3, const float* b3, int n_out,
                    float* out, hipStream_t s);

// the same on the bf16 matrix cores (pointnet_bf16.hip): conv2 / conv3 weights as bf16 hi / lo planes ([128,64], [n_out,128]);
// terms = 3 split-bf16 | 1 single-rounded (the lo planes are then not read)
int launch_pointnet_bf16(const float* pts, int n_obj, int n_points, int cin, const float* w1, const float* b1,
                         const uint16_t* w2h, const uint16_t* w2l, const float* b2, const uint16_t* w3h, const uint16_t* w3l,
                         const float* b3, int n_out, int terms, float* out, hipStream_t s);

// ---- edge cross-attention (flash style, fp32 MFMA) ----
// tiles: device array of int4 {row_base, n_tokens, q0, head}; one block per entry.
// Split-key mode (plans with too few blocks to fill the chip): tile i covers key tiles [krange[i].x, krange[i].y)
// as part krange[i].z of `parts`; partial results go to o_part [parts][rows][ldo] (un-normalised), m_part / l_part
// [parts][rows][heads]; a merge kernel writes O.  A tile with an empty key range records m = -inf, l = 0.
struct FlashSplit {
    int parts = 1;
    const int4* krange = nullptr;
    float* o_part = nullptr; float* m_part = nullptr; float* l_part = nullptr;
    size_t part_stride = 0;      // floats between parts of o_part (= rows * ldo)
    int rows = 0, heads = 0;
    int ablate = 0;              // timing experiments (garbage results): bit 0 no K/V loads after the first tile, bit 1 no LDS stores of them either
    int qg = 0;                  // half rows, head dim 64, LDS-direct kernel, no key split: 1 | 2 = 64 queries per wave (flash_attn_bf16.hip QG = 2; 2 = ORD 1), 0 = 32
    int bq = 128;                // queries per block of the tile table handed in: FLASH_BQ, or FLASH_BQ_BIG (half rows, head dim 64, LDS-direct kernel only)
};
int launch_flash_attn(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo,
                      const int4* tiles, int n_tiles, float scale_log2e, hipStream_t s, const FlashSplit* split = nullptr,
                      int head_dim = 64);      // head_dim = 512 / NUM_HEADS: 32 | 64 | 128
int launch_flash_merge(float* O, int ldo, const FlashSplit& sp, hipStream_t s, int out_split, int head_dim = 64);
// the same attention on the bf16 matrix cores (flash_attn_bf16.hip): terms = 3 split-bf16 (~1e-5) | 1 single-rounded;
// use_tr = 0 selects the gather fallback for the V operand instead of ds_read_b64_tr_b16 (tests); io_split = 1: Q (already
// scaled), K, V and O are in the split-pair format of the bf16 modes (common.h pack_split)
int launch_flash_attn_bf16(const float* Q, int ldq, const float* K, const float* V, int ldkv, float* O, int ldo,
                           const int4* tiles, int n_tiles, float scale_log2e, int terms, int use_tr, int io_split,
                           hipStream_t s, const FlashSplit* split = nullptr, int pv_terms = 3, int head_dim = 64);
bool flash_attn_bf16_supports(int head_dim, int terms, int use_tr, int io_split);   // which (head dim, format) combinations are built
constexpr int FLASH_BQ = 128;   // queries per block
constexpr int FLASH_BQ_BIG = 256;   // ... of the eight-wave variant for scenes of thousands of tokens (FlashSplit::bq)

// ---- node attention with distance bias (per scene, per head) ----
// scene_ptr: device [n_scenes+1] node offsets; bias_ptr: device [n_scenes] offsets into bias
// (layout per scene: [H][n][n], query-major).  grid covers max_n queries per scene.
// head dim dk = 512 / n_heads in {32, 64, 128}; bias may be NULL.  Also the generic (VALU) edge cross-attention for
// dk != 64 (scene_ptr = the scenes' edge ranges).
int launch_node_attn(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                     float* O, int ldo, const float* bias, const int32_t* scene_ptr,
                     const int64_t* bias_ptr, int n_scenes, int max_n, int n_heads, int dk, float scale,
                     hipStream_t s, int split_below = 1024);
// distance-bias MLP (MMG.self_attn_fc): centres = desc[:,0:3] (ld = 11)
struct DistBiasW { const float *w0, *b0, *g2, *be2, *w3, *b3, *g5, *be5, *w6, *b6; };
int launch_dist_bias(const float* desc, int ld_desc, const int32_t* scene_ptr, const int64_t* bias_ptr,
                     int n_scenes, int max_n, int n_heads, DistBiasW w, float* bias, hipStream_t s);

// ---- small fused VALU kernels ----
// edge descriptor (Gen_edge_descriptor) + conv1 of both relation encoders:
// h1[e, 0:64] = relu(W1_3d ed + b), h1[e, 64:128] = relu(W1_2d ed + b)
int launch_edge_embed(const float* desc, const int32_t* src, const int32_t* dst, int n_edges,
                      const float* w1cat /*[128,11]*/, const float* b1cat /*[128]*/, float* h1, hipStream_t s);
// x3[n, 504:512] = [desc[3:9], log desc[9], log desc[10]]
int launch_desc_tail(const float* desc, int n_nodes, float* x, int ldx, int col0, hipStream_t s);
// in-place LayerNorm over rows of `dim` (dim == 512), optional ReLU
int launch_layernorm(float* x, int ld, int rows, int dim, const float* gamma, const float* beta, int relu,
                     hipStream_t s);
// out of place; out_split = 1 writes the split-pair format of the bf16 modes (common.h pack_split)
int launch_layernorm_to(const float* x, int ld, float* y, int ldy, int rows, int dim, const float* gamma, const float* beta,
                        int relu, int out_split, hipStream_t s, const float* resid = nullptr, int ldr = 0, int r_split = 0, int x_f16 = 0);
// (x_f16: the rows of x are fp16 half rows -- what the out-projection of the single-rounded edge attention writes, GemmArgs::c_f16_cols == N)
// rowscale[m] = scale / ||x[m,:]||_2  (dim == 512)
int launch_row_invnorm(const float* x, int ld, int rows, int dim, float scale, float* out, hipStream_t s, const float* x2 = nullptr, float* out2 = nullptr);

// ---- MODEL.feature_transform glue (stn.hip): conv1 as point rows, max over an object's rows, per-object 64x64 ----
int launch_pts_conv1_rows(const float* pts, int n_obj, int P, int cin, const float* w1, const float* b1, float* rows,
                          hipStream_t s);
int launch_rowmax(const float* x, int ld, int n_obj, int P, int cols, float* out, int ldo, hipStream_t s);
int launch_apply_stn(const float* h, int ldh, const float* T, size_t rows, int P, float* out, int ldo, hipStream_t s);

// ---- 'fat' edge gate: per (edge, head) MLP 128->128->32, softmax over 32, times value ----
struct GateArgs {
    const float* kproj;      // [E, 512] head-major: kproj[e, h*64 + c]
    const float* node;       // node-side buffer, row pitch ld_node
    int ld_node;
    int gq_off;              // column offset of Gq[h*128 + o] (layer-1 node part incl. bias)
    int v_off;               // column offset of value[h*32 + m]  (head-major, see edge_gate.hip)
    const int32_t* src;      // [E]
    const int32_t* dst;      // [E]
    const float* w0k;        // [128, 64]  layer-1 weights acting on the edge half
    const float* w3;         // [32, 128]
    const float* b3;         // [32]
    float* gated;            // [E, 256]  gated[e, h*32 + m]
    float* prob;             // optional [E, 32, 8] tap (tests) or nullptr
    int n_edges;
    int use_edge = 1;        // MODEL.USE_GCN_EDGE: 0 -> the gate MLP sees the query alone (kproj / w0k unused)
    int grid_cap = 0;        // debug: persistent grid size (0 = 3 blocks per CU; vlsat_debug_option "gate_grid")
    // Fused max aggregation (Aggre_Index with GCN_AGGR = max, reference network_util.py:64-73): when `agg` is set the gated rows
    // are not stored; every wave reduces its 32 rows by source node through LDS and folds the partial maxima into
    // agg[src, h*32 + m] (row pitch ld_agg) with integer-ordered atomic max -- exact and order-independent.  `agg` must have been
    // initialised by launch_agg_init (-inf for nodes with out-edges, 0 for the others: torch_scatter's empty segment).
    float* agg = nullptr;
    int ld_agg = 0;
    int row_map = 1;         // rows of a wave: 1 = 32 edges of one head, 0 = 4 edges x 8 heads (vlsat_debug_option "gate_row_map")
};
int launch_edge_gate(const GateArgs& a, hipStream_t s, const GateArgs* twin = nullptr);     // twin: a second gate on the same edge list in the same launch (one-scene plans)
// agg[n, 0:n_ch] = rowptr[n+1] > rowptr[n] ? -inf : 0   (start values of the fused max aggregation)
int launch_agg_init(const int32_t* rowptr, int n_nodes, int n_ch, float* agg, int ld_agg, hipStream_t s, float* agg2 = nullptr);
// p[0:n] = 0 with a kernel of the library (16-byte stores when n and p allow): the forward path does not use hipMemsetAsync
int launch_zero_f32(float* p, size_t n, hipStream_t s);
// dst[r, 0:cols] = src[r, 0:cols], r < rows (pitches in floats; 16-byte accesses when sizes and pointers allow)
int launch_copy_rows(float* dst, size_t dst_ld, const float* src, size_t src_ld, int cols, size_t rows, hipStream_t s);
// any head geometry (dk query / edge channels per head, dox output channels per head): plain VALU
int launch_edge_gate_generic(const GateArgs& a, int n_heads, int dk, int dox, hipStream_t s);
// the head geometries of MODEL.NUM_HEADS in {4, 8, 16} x DIM_ATTEN in {128, 256, 512} on the fp32 matrix cores
// (edge_gate_heads.hip); returns 1 when (dk, dox) is not one of them (-> the VALU kernel)
int launch_edge_gate_heads(const GateArgs& a, int n_heads, int dk, int dox, hipStream_t s);
// ... and on the bf16 matrix cores (edge_gate_bf16_heads.hip); split-bf16 (terms = 3) not at dk = 128
bool edge_gate_bf16_heads_supports(int dk, int dox, int terms);
int launch_edge_gate_bf16_heads(const GateArgs& a, int n_heads, int dk, int dox, int terms, int kproj_split, hipStream_t s);
// the same on the bf16 matrix cores (edge_gate_bf16.hip): terms = 3 split-bf16 | 1 single-rounded; kproj_split = 1: kproj is
// in the split-pair format of the bf16 modes (common.h pack_split)
int launch_edge_gate_bf16(const GateArgs& a, int terms, int kproj_split, hipStream_t s, const GateArgs* twin = nullptr);

// ---- scatter aggregation by source node over a CSR (rowptr[N+1], order[E]) ----
// out[n, col0 + c] = reduce_{k in rowptr[n]..rowptr[n+1]} gated[order[k], c]; empty -> 0
int launch_aggregate(const float* gated, int n_ch, const int32_t* rowptr, const int32_t* order,
                     int n_nodes, int aggr, float* out, int ldo, int col0, hipStream_t s, const float* gated2 = nullptr, float* out2 = nullptr);   // gated2 / out2: a twin problem on the same graph in the same launch

// w[i] -> bf16 hi[i] + bf16 lo[i] (split-bf16 GEMM weights, one-time)
int launch_split_bf16(const float* w, size_t n, uint16_t* hi, uint16_t* lo, hipStream_t s);
int launch_to_f16(const float* w, size_t n, uint16_t* out, hipStream_t s);        // fp16 plane (precision mode fp16_mixed)

// ---- eval ranking step: softmax + top-k ranks by counting ----
int launch_softmax_rows(const float* x, int ld, int rows, int cols, float* out, int log_out, hipStream_t s);
int launch_eval_ranks(const float* obj_logits, const float* obj_probs, const float* rel, const int64_t* gt_cls,
                      const int64_t* gt_rel, const int64_t* edges, int N, int E, int C, int R, int topk_obj,
                      int topk_rel, int topk_tri, float thr, int32_t* obj_rank, int32_t* rel_rank, int32_t* tri_rank,
                      int32_t* cnt, float* sorted_probs, hipStream_t s);
// columns of the per-node sorted-probability scratch of launch_eval_ranks ([N, K] floats): only the topk largest matter
inline int eval_ranks_sorted_k(int C, int topk_tri) { return C < topk_tri ? C : topk_tri; }

// rank arrays of one batch -> += the additive counts vector of evaluate.validation (uint64 [1 + R + 2 (11 + 6 R)]; layout:
// evaluate.fields()); integer atomics only, safe from concurrent streams
int launch_eval_counts(const int32_t* obj_rank3, const int32_t* obj_rank2, const int32_t* rel_rank3, const int32_t* rel_rank2,
                       const int32_t* tri_rank3, const int32_t* tri_rank2, const int32_t* cnt, const int64_t* gt_cls,
                       const int64_t* gt_rel, const int64_t* edges, int N, int E, int R, int n_scenes, unsigned long long* out,
                       hipStream_t s);

// the additive metrics vector {scenes, N, E, four fp64 output sums, two top-1 agreement counts}; scratch: 256 * 6 doubles
int launch_scene_checksums(const float* obj3d, const float* obj2d, long N, int C, const float* rel3d, const float* rel2d, long E, int R,
                           int n_scenes, double* out9, double* scratch, hipStream_t s);

// ---- per-object input preparation ----
int launch_prepare_objects(const float* scene, const int32_t* choice, int N, int P, float* obj_points, float* desc,
                           hipStream_t s);
// *mismatches += rows of a DEVICE edge list [2,E] (int64) that differ from a plan's own (src, dst) tables, + nodes whose batch id
// run structure differs from the plan's scene partition (batch_ids may be null)
int launch_check_graph(const int64_t* edges, int64_t n_edges, const int32_t* src, const int32_t* dst, const int64_t* batch_ids,
                       int64_t n_nodes, const int32_t* scene_ptr, int n_scenes, int32_t* mismatches, hipStream_t s);
// per-object point selection (reference dataset_3dssg.py:279-289): stable per-instance index lists + n_sample draws with
// replacement from a counter-based generator (prep.hip); scratch: sample_objects_scratch_ints(n_points, n_obj) int32
size_t sample_objects_scratch_ints(int64_t n_points, int n_obj);
int launch_sample_objects(const int32_t* instances, int64_t n_points, const int32_t* ids, int n_obj, int n_sample, unsigned long long seed,
                          int32_t* id_map, int map_size, int32_t* scratch, int32_t* choice, int32_t* counts, hipStream_t s);
int launch_fc_edges(const int32_t* node_ptr, const int64_t* edge_ptr, int n_scenes, int64_t n_nodes, int64_t n_edges,
                    int64_t* edges, int64_t* batch_ids, hipStream_t s);

}  // namespace vlsat
