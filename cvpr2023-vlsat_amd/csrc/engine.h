// Internal state of libvlsat_hip.so shared by the engine translation units:
//   engine_weights.hip  handle life cycle, weight preparation (folding / hoisting / permutations), precision modes
//   engine_plan.hip     graph plan: host-side graph analysis, one device arena, asynchronous index upload
//   engine_forward.hip  the forward orchestration (vlsat_forward / vlsat_forward_train) and per-class profiling
//   engine_api.hip      kernel-level C entry points and debug hooks
// The public surface is include/vlsat.h.
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vlsat.h"
#include "common.h"
#include "kernels.h"

namespace vlsat {

enum ProfClass { PC_GEMM = 0, PC_FLASH, PC_POINTNET, PC_GATE, PC_NODE_ATTN, PC_LAYERNORM, PC_AGGREGATE, PC_MISC, PC_COUNT };

struct AttnW {          // one MultiHeadAttention block
    float *wq, *bq, *wkv, *bkv, *wo, *bo, *lng, *lnb;
    float *wqkv, *bqkv;  // self-attention: fused [3D, D]
};
struct GcnW {           // one GraphEdgeAttenNetwork block
    float *wnode, *bnode;   // [2*2D + H*(dn+de) + A, D]: Wi | Wj | Wgq | Wv
    float *we1;             // [2D, D] edge part of nn_edge.0
    float *we2, *be2;       // nn_edge.2
    float *wpe, *bpe;       // proj_edge, rows permuted head-major
    float *w0k, *w3, *b3;   // gate MLP
    float *wp0, *bp0, *wp2, *bp2;
};
struct RelHeadW { float *w1, *b1, *w2, *b2, *w3, *b3; };
// STNkd(k=64) with its five BatchNorm1d(eval) layers folded and the identity folded into the last bias
struct StnW { float *c1, *c1b, *c2, *c2b, *c3, *c3b, *f1, *f1b, *f2, *f2b, *f3, *f3b; };
// triplet_projector_2d (forward(istrain=True) only): Linear(3D, 2D) split into node-side [Wi | Wj] and edge part
struct TripletW { float *wnode = nullptr, *bnode = nullptr, *we = nullptr, *w2 = nullptr, *b2 = nullptr; };

// a pinned host staging buffer of the plan upload; `done` is recorded after the copy that reads it
struct Staging { char* p = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; };
// a workspace arena waiting for re-use; `last` (may be null) is the last forward that touched it
struct Arena { char* p = nullptr; size_t bytes = 0; hipEvent_t last = nullptr; };

}  // namespace vlsat

struct vlsat_ctx {
    VlsatDims d{};
    int dual_stream = 2;     // 2D twin stages on a second stream: 0 never, 1 launch-bound plans only (E <= 8192), 2 every plan
                             // (vlsat_debug_option "dual_stream").  A two-stream plan carries a second scratch set (NP2, Hbig2,
                             // KP2, G2, T768b, H2b: +7.3 KB per edge, 0.73 GB at the bench batch); plans whose workspace would pass
                             // DUAL_WS_BUDGET with it fall back to one stream (engine_plan.hip).  The per-class profiling keeps the
                             // two streams unless "prof_dual" is 0.
    hipStream_t side = nullptr;          // lane 1: the 2D edge chain (two-stream schedule of round 4: every 2D twin stage)
    hipStream_t side2 = nullptr;         // lane 2: the 2D node chain of the dependency-exact schedule (adapter, node cross-attention, wnode, 2D object head)
    int sched = -1;          // two-stream plans: 1 = dependency-exact three-lane schedule (round 5), 0 = the fork / join schedule of round 4,
                             // -1 = by mode and size: exact in the bf16 modes (+1.1 ... +2.5 %) on plans of more than 16 384 edges, fork / join
                             // in exact fp32, whose kernels are all matrix-pipe-bound and lose 0.4-0.8 % to the extra concurrency
                             // (profiles/r05_probes/ab_sched.txt), and on small plans (one scene per call: several replicas in flight)
                             // (vlsat_debug_option "sched"; results are bit-identical, the launches are the same)
    hipStream_t copy = nullptr;          // plan index uploads (non-blocking stream)
    std::vector<hipEvent_t> sync_ev;     // fork/join events (timing disabled), created on first use
    int fa_split = 1;        // allow the split-key edge attention for small plans (vlsat_debug_option "flash_split")
    int edge_scope = 0;      // edge cross-attention keys: 0 = the query's scene, 1 = the whole batch
    int D = 512, A = 256, H = 8, C_pt = 768;
    std::map<std::string, std::vector<float>> host;   // raw reference-layout tensors
    bool finalized = false;
    std::vector<float*> dev_allocs;
    std::vector<std::pair<const float*, size_t>> gemm_w;   // every GEMM weight matrix (pointer, elements): split eagerly in bf16 modes
    // prepared device weights
    float *pn_w1, *pn_b1, *pn_w2, *pn_b2, *pn_w3, *pn_b3;
    float *mlp_w, *mlp_b;
    float *re_w1cat, *re_b1cat;
    float *re3_w2, *re3_b2, *re3_w3, *re3_b3, *re2_w2, *re2_b2, *re2_w3, *re2_b3;
    float *ad_w1, *ad_b1, *ad_w2h, *ad_b2h;
    vlsat::DistBiasW db{};
    std::vector<vlsat::AttnW> self_attn, cross_attn, cross_rel;
    std::vector<vlsat::GcnW> gcn3, gcn2;
    vlsat::RelHeadW rel3{}, rel2{};
    vlsat::StnW stn_obj{}, stn_re3{}, stn_re2{};          // MODEL.feature_transform
    vlsat::TripletW trip{};
    float *obj3_w, *obj3_b, *obj2_w, *obj2_b;
    // profiling
    bool prof = false;
    struct Rec { int cls; hipEvent_t a, b; double flops; long kernels; };
    std::vector<Rec> recs;
    Rec open[3] = {};        // per lane (0 launch stream, 1 / 2 the side streams): interval of the kernel class being launched (see Scope)
    bool open_ok[3] = {false, false, false};
    hipEvent_t prof_base = nullptr;       // time origin of the current batch of records (intervals of the two streams are
    bool prof_base_set = false;           // put on one timeline and a class's time is the length of their UNION)
    int prof_dual = 1;                    // keep the two-stream execution while profiling (vlsat_debug_option "prof_dual")
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double acc_ms[vlsat::PC_COUNT] = {0};
    int64_t acc_n[vlsat::PC_COUNT] = {0};
    double acc_fl[vlsat::PC_COUNT] = {0};
    int debug_stop = -1;
    int gemm_no_dma = 0, gate_grid = 0, gate_row_map = 1, gate_heads_mfma = 1, flash_heads_bf16 = 1, gate_heads_bf16 = 1;      // vlsat_debug_option
    int pair_twins = 1;                      // one-scene plans (E <= pair_max_edges): the 3D / 2D twin stages as launches of two problems each (engine_forward.hip, "paired schedule"; vlsat_debug_option "pair_twins")
    int pair_max_edges = 4096;               // ... and the plan size up to which that schedule is used ("pair_max_edges")
    int gather_f16 = -1;                     // vlsat_debug_option "gather_f16": [P_i | P_j] of the node-side projection as fp16 half rows; -1 (default) = on in the single-rounding modes (prec_edge 1), 0 / 1
    int outproj_f16 = -1;                    // vlsat_debug_option "outproj_f16": the out-projection of a single-rounded edge attention writes fp16 half rows for its LayerNorm; -1 (default) = on, 0 / 1
    int gemm_k_rot = -1;                     // vlsat_debug_option "gemm_k_rot": K-tile rotation per column tile of the 8-phase GEMM; -1 (default) = 1 for half-row bf16 launches, 0 otherwise
    int gemm_no_p8 = 0;                      // vlsat_debug_option "gemm_p8": 0 keeps half-row launches off the 8-phase kernel
    int node_attn_split = 1024;              // node attention: sixteen lanes per query when the plan has fewer waves than this
    long config_epoch = 0;                   // bumped by every call that changes what a forward launches 
    int gemm_p8_part_min = 0;                // "gemm_p8_part_min": remainder tiles from which the 8-phase kernel takes them as balanced rounds (0: built-in)
    int gemm_splitk_max_tiles = 64;          // ... for launches of at most this many 64 x 64 tiles ("gemm_splitk_max_tiles"; 0: half the resident slots = the rule of
                                             // rounds 2-5, 256).  Round 6, interleaved A/B (profiles/r06_probes/ab_splitk_max_tiles*.txt): with 64 the remainder launches of the
                                             // big GEMMs (192-384 tiles) and the edge rows of a one-scene plan stay on the single-round kernel -- batch step +0.8 % bf16_mixed,
                                             // +0.7 % bf16x3, +0.3 % fp32; one-scene forwards of 21-40 objects 1.17 -> 1.10 ms; node rows of one scene (8-52 tiles) keep split-K
    int gemm_splitk = 1;                     // small GEMM launches take the split-K kernel (gemm_splitk.hip)
    float* sk_ws[3] = {nullptr, nullptr, nullptr};    // its workspace + counters, one set per lane: [0] launch stream, [1] / [2] the side streams
    unsigned* sk_cnt[3] = {nullptr, nullptr, nullptr};
    int flash_bq_big_min = 4096;             // ... and the smallest scene (edges) from which a plan gets them (debug option "flash_bq_big_min": plans created afterwards)
    int flash_qg = 0;                        // 64 queries per wave in the half-row bf16 edge attention (debug option "flash_qg" 1 | 2; FlashSplit::qg)
    int flash_bq_big = 1;                    // 256-query tiles for plans whose scenes all have >= 4096 edges (debug option "flash_bq_big" 0: always 128)
    int flash_dma = 1;                       // half-row bf16 edge attention: K / V by LDS-direct loads, one tile ahead (0: register-staged, round 3; 3 | 4: rings of three / four tile buffers)
    int gate_fuse_agg = 1;                   // gate at the default head geometry, GCN_AGGR = max: aggregation fused into the gate kernel: 0 never, 1 the bf16 modes, 2 fp32 as well ("gate_fuse_agg")
    int flash_ablate = 0;                    // timing experiments on the bf16 edge attention (FlashSplit::ablate; results are garbage)
    int flash_pv_terms = 3;                  // split-bf16 edge attention: MFMAs per P.V product (3, or 2 = probabilities single-rounded)
    int ln_resid = 1;                        // split-pair mode: post-attention residual added in the LayerNorm kernel
    int half_fmt = 1;                        // single-rounding modes: those tensors as plain bf16 (half rows) instead of split pairs
    int split_fmt = 1;                       // bf16 modes: edge tensors between matrix kernels in the split-pair format
    int pointnet_bf16 = 1, gate_bf16 = 1;    // bf16 modes: object encoder / edge gate on the bf16 matrix cores
    int flash_bf16 = 1, flash_tr = 1;        // bf16 modes: attention on the bf16 matrix cores / V operand by LDS transpose read
    long gemm_launches = 0;  // kernels launched by launch_gemm for this handle (main + tail launches)
    int cur_N = -1;          // node count of the plan whose forward is being enqueued (row class of a GEMM launch)
    // GEMM operand precision: 0 exact fp32 MFMA, 1 bf16, 3 split-bf16 (vlsat_set_gemm_precision); prec_edge / prec_node
    // are what the edge-row / node-row launches actually use (mode 2 = mixed: bf16 on edge rows, bf16x3 on node rows)
    int prec = 0, prec_edge = 0, prec_node = 0;
    int half_f16 = 0;        // mode 5 ("fp16_mixed"): mode 2 with fp16 in the half rows and on the matrix cores of the edge-row kernels (v_mfma_f32_32x32x16_f16)
    int prec_attn = 0;       // ... and what the edge cross-attention (its three projections and the attention kernel) uses: prec_edge, except in
                             // mode 4 = split-bf16 with a single-rounded edge attention (1)
    std::map<const float*, std::pair<uint16_t*, uint16_t*>> split;
    std::map<const float*, uint16_t*> f16w;        // fp16 planes of the same matrices (made when mode 5 is set)
    // workspace arenas of destroyed plans, re-used by the next plan that fits (an eval loop may build one plan per
    // scene; hipMalloc/hipFree per scene would dominate small scenes), pinned upload buffers, spare events
    std::vector<vlsat::Arena> arena_pool;
    std::vector<vlsat::Arena> arena_trash;          // too many pooled: freed once their last forward has completed
    std::vector<vlsat::Staging> staging;
    std::vector<hipEvent_t> spare_ev;
};

struct vlsat_plan_s {
    vlsat_ctx* h = nullptr;
    int64_t N = 0, E = 0;
    int P = 0, S = 0, max_n = 0, is_fc = 0;
    std::vector<int32_t> node_ptr;          // [S+1]
    std::vector<int64_t> edge_ptr;          // [S+1]
    size_t ws_bytes = 0;
    char* arena = nullptr;
    size_t arena_bytes = 0;
    hipEvent_t uploaded = nullptr;          // index tables are in place (recorded on the handle's copy stream)
    bool upload_pending = false;            // the next forward must make its stream wait for `uploaded`
    hipEvent_t last_use = nullptr;          // recorded at the end of every forward on that forward's stream
    bool used = false;
    // device index arrays
    int32_t *d_src, *d_dst, *d_rowptr, *d_order, *d_scene_ptr;
    int32_t* d_edge_ptr32 = nullptr;        // [S+1] edge ranges of the scenes (generic edge attention for d_k != 64)
    int max_e = 0;                          // most edges in one scene
    int64_t* d_bias_ptr;
    int4* d_tiles;
    int n_tiles = 0;
    int4* d_tiles_big = nullptr;             // the same attention cut into FLASH_BQ_BIG-query tiles: only when every scene has >= 4096 edges (engine_plan.hip)
    int n_tiles_big = 0;
    // split-key mode of the edge attention for plans with too few blocks to fill the chip (flash_attn_*.hip)
    int fa_parts = 1;
    int4* d_krange = nullptr;
    float *fa_opart = nullptr, *fa_m = nullptr, *fa_l = nullptr;
    double flash_flops = 0;
    // device float buffers
    float *F, *X3, *X2, *NP, *QKVn, *On, *T256, *T768, *rs, *bias;
    float *H1, *H2, *E3, *E2, *Hbig, *KP, *G, *Qe, *KVe, *Oe, *R1, *R2, *prob;
    // Small plans (launch-bound: one scene per call) run the 2D twin of every stage -- relation encoder, adapter,
    // gcn_2ds, the query projection of the edge attention, the 2D heads -- on a second stream, concurrently with
    // the 3D twin.  The twins never touch each other's tensors; they only shared scratch, so the 2D side gets its own.
    float* stn_ws = nullptr;                        // MODEL.feature_transform scratch (carved per phase in stn_encoder)
    size_t stn_ws_floats = 0;
    bool dual = false;
    float *NP2 = nullptr, *Hbig2 = nullptr, *KP2 = nullptr, *G2 = nullptr, *T768b = nullptr, *rs2 = nullptr, *H2b = nullptr;
    // node cross-attention: its query / output rows (the self-attention's QKVn / On may be in use on another lane) and the
    // key | value projection of X3 [kvx_slots][N, 2 D], computed on the 3D lane right after the self-attention (so that gcn_3ds may
    // overwrite X3 without waiting for the 2D side); two-stream plans keep one slot per layer
    float *Q2n = nullptr, *On2 = nullptr, *KVx = nullptr;
    int kvx_slots = 1;
    // scratch of vlsat_process_val_counts: the four outputs and the two object-probability tables (floats), the rank tables (int32)
    float* ev_f = nullptr;
    int32_t* ev_i = nullptr;
    float* KVe2 = nullptr;                  // second K|V buffer of the edge cross-attention (two-stream plans: the 3D lane runs a layer ahead)
};

namespace vlsat {

extern const char* kProfNames[PC_COUNT];

// scratch buffers of one modality branch
struct Scratch { float *NP, *Hbig, *KP, *G, *T768, *R1, *R2, *rs, *H2; };

// row pitches that depend on MODEL.DIM_ATTEN (A): node features carry [x (512) | aggregated message (A)], the node-side
// projection buffer [P_i 1024 | P_j 1024 | Gq H*(2 d_k) = 1024 | value A]
inline int ldx_of(const vlsat_ctx* h) { return h->D + h->A; }
inline int npc_of(const vlsat_ctx* h) { return 6 * h->D + h->A; }
// [P_i | P_j] as fp16 half rows (written by gcn_node_project, gathered by nn_edge.0): the single-rounding modes unless "gather_f16" says otherwise;
// never in the exact-fp32 mode and never when the node rows run in fp32
inline bool gather_f16_on(const vlsat_ctx* h) { return h->prec_edge != 0 && h->prec_node != 0 && (h->gather_f16 >= 0 ? h->gather_f16 != 0 : h->prec_edge == 1); }
inline bool default_heads(const vlsat_ctx* h) { return h->H == 8 && h->A == 256; }
// a plan whose workspace would exceed this with the second scratch set of the two-stream mode runs on one stream
constexpr size_t DUAL_WS_BUDGET = size_t(48) << 30;

#define RUN(expr)                  \
    do {                           \
        int _r = (expr);           \
        if (_r) return _r;         \
    } while (0)

// engine_weights.hip
const char* last_error_cstr();
int split_all_weights(vlsat_ctx* h);
// engine_plan.hip
hipEvent_t take_event(vlsat_ctx* h);
void give_event(vlsat_ctx* h, hipEvent_t e);
void release_plan_resources(vlsat_ctx* h);       // frees pools (vlsat_destroy)

}  // namespace vlsat
