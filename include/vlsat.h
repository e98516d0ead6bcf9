/*
 * vlsat.h -- C ABI of libvlsat_hip.so: the MI355X (gfx950) implementation of VL-SAT's
 * per-scene eval forward.
 *
 * The reference has NO native/FFI layer (pure Python on PyTorch + PyG); its boundary for this
 * path is the nn.Module call
 *     Mmgnet.forward(obj_points, obj_2d_feats, edge_indices, descriptor, batch_ids, istrain=False)
 *         reference src/model/SGFN_MMG/model.py:288-335, called from process_val :458-460,
 *         which MMGNet.validation calls at src/model/model.py:203-211.
 * Every entry point below replaces a piece of that call; the reference line it stands for is
 * cited on each declaration.  The ctypes binding a reference maintainer would add is shown in
 * INTEGRATION.md and implemented in cvpr2023-vlsat_amd/lib.py.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device addresses
 *     (torch: tensor.data_ptr()); `stream` is a hipStream_t passed as void*
 *     (torch: torch.cuda.current_stream().cuda_stream); NULL = the default stream.
 *   - all tensors fp32 row-major contiguous unless noted; indices int64 like the reference.
 *   - every function returns 0 on success or a negative VLSAT_E* code; it never throws and
 *     never calls exit(); vlsat_last_error() returns a thread-local message for the last failure.
 *   - vlsat_forward and the vlsat_k_* kernels are asynchronous on `stream` and do not synchronise.
 *   - vlsat_plan_create / vlsat_plan_destroy never wait for the device either: the index tables are uploaded
 *     asynchronously from pinned memory and the first forward of the plan waits for them ON THE DEVICE; a destroyed
 *     plan's workspace is recycled behind the event of its last forward.  Device-wide waits exist only in
 *     vlsat_load_weight (when it replaces an already finalised set), vlsat_set_gemm_precision, vlsat_destroy and the
 *     vlsat_debug_* readers.
 *   - a handle (weights) may be shared by several plans; a plan owns its workspace and is NOT
 *     re-entrant (one forward at a time per plan; ), mirroring one nn.Module instance; a handle is driven from one
 *     host thread at a time, and its forwards are ordered on ONE stream at a time (small scratch buffers -- the split-K
 *     workspace of small GEMM launches -- belong to the handle: moving to another stream needs an event / sync between
 *     the last forward on the old stream and the first on the new one; separate handles are independent).
 *   - several handles may be driven from several host threads at the same time (one thread and one stream per handle:
 *     evaluate.validation(workers=K)).  The forward path launches only kernels of this library -- no hipMemsetAsync, no
 *     runtime device-to-device copy: a hipMemsetAsync issued from several threads at once was caught writing a foreign
 *     pattern about once per 20 000 calls.  A host that drives handles from threads should keep
 *     runtime fills (and library reductions that use them internally) out of those threads as well.
 */
#ifndef VLSAT_H
#define VLSAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLSAT_OK            0
#define VLSAT_EINVAL       (-1)   /* bad argument / shape */
#define VLSAT_EHIP         (-2)   /* HIP runtime error (message has hipGetErrorString) */
#define VLSAT_ESTATE       (-3)   /* call order (weights missing / not finalised) */
#define VLSAT_EGRAPH       (-4)   /* graph layout not supported as given (see vlsat_plan_create) */
#define VLSAT_ENOMEM       (-5)

typedef struct vlsat_ctx*  vlsat_handle;
typedef struct vlsat_plan_s* vlsat_plan;

/* Hyper-parameters = the MODEL keys Mmgnet.__init__ reads (reference SGFN_MMG/model.py:26-130,
 * config/mmgnet.json:26-58). */
typedef struct {
    int32_t n_layers;        /* MODEL.N_LAYERS */
    int32_t n_heads;         /* MODEL.NUM_HEADS (8); built: 4, 8, 16 */
    int32_t dim_atten;       /* MODEL.DIM_ATTEN (256); built: 128, 256, 512 */
    int32_t gcn_aggr;        /* MODEL.GCN_AGGR: 0 max, 1 add, 2 mean */
    int32_t dim_point;       /* 3, +3 with MODEL.USE_RGB, +3 with MODEL.USE_NORMAL (SGFN_MMG/model.py:31-35) */
    int32_t n_obj_class;     /* 160 */
    int32_t n_rel_class;     /* 26 (27 in the single-label setting) */
    float   obj_logit_scale; /* log(1/0.07): never checkpointed by the reference */
    int32_t use_gcn_edge;    /* MODEL.USE_GCN_EDGE (1): gate MLP on cat[q,k]; 0: on q alone (network_MMG.py:72-75) */
    int32_t multi_rel_outputs; /* MODEL.multi_rel_outputs (1): sigmoid relation head; 0: log_softmax (SGFN_MMG/model.py:113-130) */
    int32_t feature_transform; /* MODEL.feature_transform (0): STNkd 64x64 transform after conv1 of the three encoders */
    /* MODEL.WITH_BN needs no field: the relation heads' bn1/bn2 tensors are folded when the checkpoint has them */
} VlsatDims;

const char* vlsat_last_error(void);
/* library / build identification, e.g. "vlsat-hip gfx950 r4 (fp32-mfma | bf16x3 | bf16_mixed | bf16)" */
const char* vlsat_version(void);

/* Mmgnet.__init__ (module construction), reference SGFN_MMG/model.py:20-159. */
int vlsat_create(const VlsatDims* dims, vlsat_handle* out);
void vlsat_destroy(vlsat_handle h);

/* BaseModel.load / load_state_dict, reference model_utils/model_base.py:75-129: `name` is the
 * reference state_dict key prefixed by the sub-module name ("mmg.gcn_3ds.0.edgeatten.nn_edge.0.weight");
 * `host` is fp32, `count` elements.  Unknown names -> VLSAT_EINVAL.  Like BaseModel.load, loading may be repeated:
 * the first call after a vlsat_finalize_weights starts a new, complete set (it waits for the device to go idle and
 * drops the previous device tensors); existing plans stay valid.  "triplet_projector_2d.{0,3}.{weight,bias}" are
 * optional and only read by vlsat_forward_train. */
int vlsat_load_weight(vlsat_handle h, const char* name, const float* host, size_t count);
/* Folds BatchNorm(eval), splits/concatenates/permutes the projections the kernels use and
 * uploads everything to the device.  Fails with VLSAT_ESTATE listing the first missing tensor. */
int vlsat_finalize_weights(vlsat_handle h);

/* Graph analysis for one batch = what MMG.forward derives from batch_ids each call
 * (reference network_MMG.py:183-205) plus the PyG gather/scatter index bookkeeping
 * (network_util.py:50-73).  HOST pointers: batch_ids [N] int64 (scene id per node, scenes
 * contiguous and ascending), edges [2,E] int64 (row 0 = source, row 1 = target, already offset
 * like collate_fn_mmg does, reference DataLoader.py:167-172).  Requirements: every edge joins two
 * nodes of one scene; edges of a scene are contiguous and scenes appear in node order
 * (otherwise VLSAT_EGRAPH: the Python glue then permutes edges and retries).  Within a scene the
 * edge order is arbitrary.  Allocates (or recycles) the per-plan device workspace for (N, E, P).  Returns without
 * waiting for the device; the host arrays may be freed as soon as it returns. */
int vlsat_plan_create(vlsat_handle h, const int64_t* batch_ids_host, const int64_t* edges_host,
                      int64_t n_nodes, int64_t n_edges, int32_t n_points, vlsat_plan* out);
void vlsat_plan_destroy(vlsat_plan p);
/* n_scenes, workspace bytes, and whether the fully-connected fast paths were selected */
int vlsat_plan_info(vlsat_plan p, int32_t* n_scenes, size_t* workspace_bytes, int32_t* is_fc);

/* Does a DEVICE copy of a graph equal what this plan was built from?  *mismatches (device int32, zeroed by the caller) receives
 * the number of columns of edges_dev ([2,E] int64, the layout Mmgnet.forward takes) that differ from the plan's edge list plus the
 * nodes at which batch_ids_dev ([N] int64, may be NULL) starts a new run where the plan has no scene boundary (or the reverse).
 * Asynchronous on `stream`; no host wait.  For hosts that name a graph by a key (e.g. the object counts of fully-connected
 * scenes) instead of handing the edge list over: one call per new key makes the key's claim checked -- a permuted edge list
 * would otherwise attribute every relation row to the wrong edge (reference edge order: src/dataset/dataset_3dssg.py:264-266). */
int vlsat_plan_check_graph(vlsat_plan p, const int64_t* edges_dev, const int64_t* batch_ids_dev, int32_t* mismatches, void* stream);

/* Mmgnet.forward(..., istrain=False), reference SGFN_MMG/model.py:288-335.
 * Device pointers: obj_points [N,dim_point,P], obj_2d_feats [N,512], descriptor [N,11];
 * outputs obj_logits_3d/2d [N,n_obj_class] (logits x exp(scale)), rel_cls_3d/2d
 * [E,n_rel_class] (post-sigmoid; log_softmax when multi_rel_outputs = 0).  Edge order of the outputs = edge
 * order given to the plan.
 * 3D-only mode: pass NULL for BOTH obj_logits_2d and rel_cls_2d (obj_2d_feats may then be NULL too):
 * the whole 2D branch is skipped -- exact, because the 3D branch never reads 2D tensors. */
int vlsat_forward(vlsat_handle h, vlsat_plan p,
                  const float* obj_points, const float* obj_2d_feats, const float* descriptor,
                  float* obj_logits_3d, float* obj_logits_2d, float* rel_cls_3d, float* rel_cls_2d,
                  void* stream);

/* Mmgnet.forward(..., istrain=True) in eval mode (no dropout, no autograd), reference SGFN_MMG/model.py:291-292,312,
 * 319-322,332-333: the four outputs of vlsat_forward plus obj_feature_3d_mimic [N,512] (first 512 columns of the
 * point encoder's output), obj_features_2d_mimic [N,512] (the adapter's output) and gcn_edge_feature_2d_dis [E,512]
 * (triplet_projector_2d on cat[x2[ei[0]], x2[ei[1]], e2]); the tuple's eighth element is exp(dims.obj_logit_scale).
 * Needs the triplet_projector_2d weights. */
int vlsat_forward_train(vlsat_handle h, vlsat_plan p,
                        const float* obj_points, const float* obj_2d_feats, const float* descriptor,
                        float* obj_logits_3d, float* obj_logits_2d, float* rel_cls_3d, float* rel_cls_2d,
                        float* obj_feature_3d_mimic, float* obj_features_2d_mimic, float* gcn_edge_feature_2d_dis,
                        void* stream);


/* Operand precision of the matrix kernels inside vlsat_forward (BASELINE configs[2], "bf16 MFMA for the
 * QKV/FFN GEMMs"): 0 = exact fp32 MFMA (default; BASELINE configs[1]); 3 = split-bf16 (a_hi.w_hi + a_lo.w_hi +
 * a_hi.w_lo on v_mfma_f32_32x32x16_bf16, fp32 accumulate, ~1e-5 error); 1 = single-rounded bf16 operands
 * everywhere (~2e-2 on the x14.29 object logits: outside the config's 1e-2); 2 = mixed: single-rounded bf16 on the
 * edge-row GEMMs / attention / gate and split-bf16 on the node-row GEMMs (meets 1e-2 at Xavier-scale weights);
 * 4 = split-bf16 everywhere except the edge cross-attention (reference network_MMG.py:228-234: its q / k|v / out projections and the
 * attention itself), which is single-rounded -- the 3D outputs never read that block and keep mode 3's accuracy, the 2D outputs hold
 * 1e-2 on weights where mode 2 does not;
 * 5 = mode 2 on FP16: the half-row tensors between the edge-row kernels hold fp16 instead of bf16 and those kernels (8-phase / ring / small
 * GEMMs, edge attention, gate, PointNet) run v_mfma_f32_32x32x16_f16 -- the bf16 rate on CDNA4, 2^-12 instead of 2^-9 per stored value and
 * operand: 7e-4 where mode 2 has 5e-3, 3-4 % slower (the chip is power-limited and fp16 multipliers draw more); values beyond +-65504 saturate.
 * Activations in HBM, softmax and LayerNorm stay fp32 in every mode.  May be changed between forwards; the bf16
 * planes of the weights are made inside this call (it waits for the device), never inside a forward. */
int vlsat_set_gemm_precision(vlsat_handle h, int32_t mode);

/* Which 3D edges a 2D edge attends to in the edge cross-attention (reference network_MMG.py:228-234), for plans
 * created AFTER the call.  0 (default): the edges of its own scene -- validation()'s contract, which runs one scene
 * per call (src/model/model.py:185).  1: every edge of the batch -- what the reference computes when a single call
 * carries several scenes, because that attention has no scene mask; the 3D outputs are identical in
 * both modes, the 2D outputs are not. */
int vlsat_set_edge_attention_scope(vlsat_handle h, int32_t scope);

/* Per-kernel timing of the forward with HIP events on `stream` (bench.py roofline leg).
 * enable=1: every launch of every kernel class is bracketed by hipEventRecord on the launch
 * stream; vlsat_profile_read synchronises, accumulates and returns per-class totals since
 * the last read.  Classes are enumerated by vlsat_profile_class_name(i), i in [0, n). */
int vlsat_profile_enable(vlsat_handle h, int32_t enable);
int vlsat_profile_num_classes(void);
const char* vlsat_profile_class_name(int32_t cls);
int vlsat_profile_read(vlsat_handle h, int32_t cls, double* total_ms, int64_t* launches, double* flops);

/* -------- single-kernel entry points (unit/parity tests; all device pointers) ---------------- */

/* C[M,N] = act(rowscale[m] * (reluA?(A)[M,K] . W[N,K]^T) + bias[n] + resid_scale*resid[m,n]
 *              + g0[gi0[m], n] + g1[gi1[m], n]);   act: 0 none, 1 relu, 2 sigmoid.
 * Replaces every nn.Linear / Conv1d(k=1) call site of the path (SURVEY §2 "addmm" row).
 * K % 32 == 0; lda/ldw % 4 == 0; any pointer except A, W, C may be NULL; rowscale may not be
 * combined with resid/g0/g1 (VLSAT_EINVAL). */
int vlsat_k_gemm(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc,
                 int32_t M, int32_t N, int32_t K,
                 const float* bias, const float* rowscale,
                 const float* resid, int32_t ldr, float resid_scale,
                 const float* g0, const int32_t* gi0, int32_t ldg0,
                 const float* g1, const int32_t* gi1, int32_t ldg1,
                 int32_t relu_a, int32_t act, void* stream);

/* Weight planes of the bf16 modes: w[i] ~= bf16 hi[i] + bf16 lo[i] (asynchronous on `stream`). */
int vlsat_k_split_bf16(const float* w, size_t n, uint16_t* hi, uint16_t* lo, void* stream);

/* vlsat_k_gemm on the bf16 matrix cores (BASELINE configs[2]) with caller-provided planes of W (dense [N,K], ldw = K):
 * prec 1 = single-rounded bf16 operands, 3 = split-bf16 (three MFMAs per product).  A stays fp32 and is split inside
 * the kernel.  no_dma = 1: the VGPR-staged operand pipe of round 1 instead of the LDS-direct one; prefetch: slices of
 * look-ahead of the A-panel prefetch (0 = off, -1 = default).  fmt: bit 0 = A, bit 1 = resid, bit 2 = C are in the
 * split-pair format of the split-bf16 mode (one 32-bit word per element: bf16 hi = rne(x) in the upper half, bf16 lo =
 * rne(x - hi) in the lower half) or, with bit 5 set, in the half-row format of the single-rounding modes (bf16 values
 * at byte 2 * column of the fp32-pitched row; prec 1 only); bit 3: g0 / g1 are FP16 half rows (fp16 values at byte 2 * column of the
 * fp32-pitched table row; no residual, N % 256 == 0), bits 25..28: k = that many times 256 leading columns of C are written as fp16 half
 * rows, clamped to +-65504 (the node-side projection's [P_i | P_j], which nn_edge.0 gathers per edge); bits 4 / 6 / 7 are benchmarking switches (
 * no ring kernel, small launches on the split-K kernel, ring kernel with 128 x 256 tiles; bits 10 / 11: its
 * 32-wide k slices / single fragment set in the half-row mode) and bits 8 / 9 timing
 * ablations (no operand loads / no MFMAs: garbage results); c_scale multiplies C
 * last.  Asynchronous. */
int vlsat_k_gemm_planes(const float* A, int32_t lda, const float* W, const uint16_t* Whi, const uint16_t* Wlo, int32_t ldw,
                        float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, const float* bias,
                        const float* resid, int32_t ldr, float resid_scale,
                        const float* g0, const int32_t* gi0, int32_t ldg0,
                        const float* g1, const int32_t* gi1, int32_t ldg1,
                        int32_t relu_a, int32_t act, int32_t prec, int32_t no_dma, int32_t prefetch, int32_t fmt,
                        float c_scale, void* stream);
/* Test entry point: the same with the planes made on the spot from W (allocates, synchronises). */
int vlsat_k_gemm_bf16(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc,
                      int32_t M, int32_t N, int32_t K, const float* bias,
                      const float* resid, int32_t ldr, float resid_scale,
                      const float* g0, const int32_t* gi0, int32_t ldg0,
                      const float* g1, const int32_t* gi1, int32_t ldg1,
                      int32_t relu_a, int32_t act, int32_t prec, int32_t no_dma, void* stream);

/* PointNetfeat.forward (obj_encoder), reference network_PointNet.py:141-164:
 * pts [N,3,P] -> out [N,768] = max_p relu(W3 relu(W2 relu(W1 x + b1) + b2) + b3).
 * Weights in the reference layout (W1 [64,3], W2 [128,64], W3 [768,128]). */
int vlsat_k_pointnet(const float* pts, int32_t n_obj, int32_t n_points,
                     const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* w3, const float* b3, int32_t n_out, float* out, void* stream);

/* ScaledDotProductAttention core for the edge cross-attention (reference attention.py:60-76 as
 * called from network_MMG.py:231): per scene s and head h, O = softmax(Q K^T * scale) V with
 * Q,K,V,O [T,512] (head h = columns 64h..64h+63), tokens of scene s = rows tok_ptr[s]..tok_ptr[s+1].
 * tok_ptr is a HOST array of n_scenes+1 int64.  Never materialises the T x T scores. */
int vlsat_k_flash_attn(const float* Q, const float* K, const float* V, float* O,
                       int32_t ld, const int64_t* tok_ptr_host, int32_t n_scenes, int32_t n_heads,
                       float scale, void* stream);

/* The same attention on the bf16 matrix cores (BASELINE configs[2]): terms = 3 split-bf16 (three MFMAs per product,
 * ~1e-5) or 1 (single-rounded operands); use_tr = 1 reads the V operand with the LDS transpose read, 0 gathers it,
 * 2 = transpose read with Q (pre-multiplied by scale*log2 e), K, V and O in the split-pair format, 3 = the same in the
 * half-row format (terms = 1). */
int vlsat_k_flash_attn_bf16(const float* Q, const float* K, const float* V, float* O,
                            int32_t ld, const int64_t* tok_ptr_host, int32_t n_scenes, int32_t n_heads,
                            float scale, int32_t terms, int32_t use_tr, void* stream);

/* LayerNorm over rows of 512 in place, optional ReLU (reference attention.py:122 + MMG :236-248). */
int vlsat_k_layernorm(float* x, int32_t ld, int32_t rows, int32_t dim, const float* gamma,
                      const float* beta, int32_t relu, void* stream);

/* The 'fat' gate of MultiHeadedEdgeAttention.forward (reference network_MMG.py:96-104): per (edge e, head h)
 *     hidden = relu(Gq[src[e], h, :] + W0k . kproj[e, h, :]);  prob = softmax(W3 . hidden + b3);  gated = prob * value[dst[e], h, :]
 * on the operands the forward prepares: kproj [E, H*dk] = proj_edge output HEAD-MAJOR (column h*dk + c
 * is the reference's k.view(E,dk,H)[e,c,h]); `node` = the node-side buffer with row pitch ld_node holding, at column gq_off,
 * Gq[n, h*2dk + o] = b0[o] + sum_c W0[o,c] q.view(N,dk,H)[n,c,h] (the query half of nn.0 applied per node, bias included)
 * and, at column v_off, value[n, h*dox + m] = proj_value output head-major; src / dst [E] int32 (ei[0] / ei[1]);
 * w0k [2dk, dk] = nn.0.weight[:, dk:2dk], w3 [dox, 2dk] = nn.3.weight, b3 [dox].  Outputs: gated [E, H*dox] head-major
 * (column h*dox + m; the reference's is m*H + h) and, if not NULL, prob [E, dox, H] in the REFERENCE layout.
 * use_edge = 0: MODEL.USE_GCN_EDGE false (hidden = relu(Gq), kproj / w0k unread).  ld_node, gq_off, v_off % 4 == 0.
 * variant: 0 = the kernel the fp32 forward picks for the geometry, 1 = VALU kernel (any geometry), 2 = fp32 MFMA template
 * (NUM_HEADS in {4,8,16} x DIM_ATTEN in {128,256,512}), 3 / 4 = bf16 matrix cores, split-bf16 / single-rounded (kproj fp32).
 * All device pointers; asynchronous. */
int vlsat_k_edge_gate(const float* kproj, const float* node, int32_t ld_node, int32_t gq_off, int32_t v_off,
                      const int32_t* src, const int32_t* dst, const float* w0k, const float* w3, const float* b3,
                      float* gated, float* prob, int32_t n_edges, int32_t n_heads, int32_t dk, int32_t dox,
                      int32_t use_edge, int32_t variant, void* stream);

/* Aggre_Index (reference network_util.py:64-73, torch_scatter semantics): out[n, col0 + c] = reduce over the rows e of
 * gated [E, n_ch] (dense) with index[e] == n; aggr 0 max | 1 add | 2 mean (MODEL.GCN_AGGR); empty segment -> 0.
 * index_host: HOST int64 [E] (ei[0] under flow = target_to_source).  Deterministic (CSR, no atomics).  Synchronises. */
int vlsat_k_aggregate(const float* gated, int32_t n_ch, const int64_t* index_host, int64_t n_edges, int32_t n_nodes,
                      int32_t aggr, float* out, int32_t ldo, int32_t col0, void* stream);

/* ScaledDotProductAttention core of the node self / cross attention (reference attention.py:60-76 as called from
 * network_MMG.py:217-218) with the additive distance bias and the block-diagonal scene mask of network_MMG.py:188-203:
 * per scene s, head h: O = softmax(scale * Q K^T + bias[s,h]) V over the scene's nodes; head h = columns h*dk.. of Q/K/V/O
 * with dk = 512 / n_heads in {32, 64, 128}.  node_ptr_host: HOST int64 [n_scenes+1]; bias (may be NULL): scene s at float
 * offset sum_{t<s} H n_t^2, layout [H][n_s][n_s] (query-major) -- what vlsat_k_dist_bias writes.  lanes_per_query: 0 = the
 * forward's choice, 1 / 16 = force the one-query-per-lane / sixteen-lanes-per-query kernel.  Synchronises. */
int vlsat_k_node_attn(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv,
                      float* O, int32_t ldo, const float* bias, const int64_t* node_ptr_host, int32_t n_scenes,
                      int32_t n_heads, float scale, int32_t lanes_per_query, void* stream);

/* The distance-bias MLP MMG.forward runs once per call (reference network_MMG.py:190-203, self_attn_fc :165-173):
 * bias[s][h][a][b] = Linear(32,H)(LN(relu(Linear(32,32)(LN(relu(Linear(4,32)([c_b - c_a, |c_b - c_a|]))))))) for the nodes a, b
 * of scene s; c = desc[:, 0:3] (row pitch ld_desc).  Weights in the reference layout (self_attn_fc.{0,2,3,5,6}).
 * Output layout as vlsat_k_node_attn reads it.  Synchronises. */
int vlsat_k_dist_bias(const float* desc, int32_t ld_desc, const int64_t* node_ptr_host, int32_t n_scenes, int32_t n_heads,
                      const float* w0, const float* b0, const float* ln2_w, const float* ln2_b, const float* w3,
                      const float* b3, const float* ln5_w, const float* ln5_b, const float* w6, const float* b6,
                      float* bias, void* stream);

/* -------- input preparation (the step BEFORE the path: the data loader, reference dataset_3dssg.py:279-294) --- */

/* Per object n: gather P sampled points scene_points[choice[n, :]] ([Npts,3] fp32, choice int32 [N,P]),
 * descriptor[n] = gen_descriptor of them (reference src/utils/op_utils.py:47-64: centroid, unbiased std,
 * max-min, volume, max dim), obj_points[n] = the points minus their mean, laid out [N,3,P]
 * (zero_mean dataset_3dssg.py:189-191 + the permute of src/model/model.py:79).  All device pointers. */
int vlsat_prepare_objects(const float* scene_points, const int32_t* choice, int32_t n_obj, int32_t n_points,
                          float* obj_points, float* descriptor, void* stream);

/* Per-object point selection on the device -- the step in front of vlsat_prepare_objects (reference
 * src/dataset/dataset_3dssg.py:279-289: obj_pointset = points[np.where(instances == id)[0]]; choice = np.random.choice(len, P,
 * replace=True)).  instances int32 [Npts] (instance id of every scene point), instance_ids int32 [N] (the scene's objects, distinct,
 * each < map_size).  Builds every instance's point indices in ascending order (np.where's) by a stable segmented compaction and
 * draws n_sample of them with replacement from a counter-based generator: draw (obj, j) is a function of (seed, obj * n_sample + j)
 * only -- splitmix64 of seed + 0x9E3779B97F4A7C15 (obj n_sample + j + 1), top 32 bits scaled to the instance's point count
 * (csrc/prep.hip; oracle/prep_oracle.py restates it bit for bit; it is NOT numpy's Mersenne stream).  choice int32 [N, n_sample]
 * feeds vlsat_prepare_objects; counts int32 [N] = points per instance (0: the instance does not occur; its choices are 0).
 * id_map: int32 [map_size] scratch; scratch: int32 [vlsat_sample_objects_scratch(n_points, n_obj)].  Device pointers; asynchronous. */
size_t vlsat_sample_objects_scratch(int64_t n_points, int32_t n_obj);
int vlsat_sample_objects(const int32_t* instances, int64_t n_points, const int32_t* instance_ids, int32_t n_obj, int32_t n_sample,
                         uint64_t seed, int32_t* id_map, int32_t map_size, int32_t* scratch, int32_t* choice, int32_t* counts,
                         void* stream);

/* Fully-connected directed edges without self loops, source-major, for a batch of scenes with node
 * offsets applied, and the batch ids (dataset_3dssg.py:264-266 + collate_fn_mmg DataLoader.py:160-172).
 * node_ptr int32 [S+1] and edge_ptr int64 [S+1] (edge_ptr[s+1]-edge_ptr[s] = n_s(n_s-1)) are device arrays;
 * edges is [2,E] int64 (the layout Mmgnet.forward takes), batch_ids [N] int64. */
int vlsat_fc_edges(const int32_t* node_ptr, const int64_t* edge_ptr, int32_t n_scenes, int64_t n_nodes, int64_t n_edges,
                   int64_t* edges, int64_t* batch_ids, void* stream);

/* -------- eval ranking step (the caller of the path: process_val, reference SGFN_MMG/model.py:463-472) --- */

/* out[r, :] = softmax(x[r, 0:cols]) -- F.softmax(objs_pred) of evaluate_triplet_topk
 * (reference src/utils/eva_utils_acc.py:144).  out is dense [rows, cols]. */
int vlsat_k_softmax_rows(const float* x, int32_t ld, int32_t rows, int32_t cols, float* out, void* stream);

/* evaluate_topk_object (eva_utils_acc.py:27-39), evaluate_topk_predicate (:42-79) and
 * evaluate_triplet_topk (:137-213, multi_rel_outputs=True, use_clip=True) as counting kernels; all
 * device pointers.  gt_rel is the multi-hot [E,R] int64 target, edges is [E,2] (from, to) int64 like
 * the `edge_indices` the reference passes.  Outputs: obj_rank [N]; rel_rank / tri_rank [E,R] hold, for
 * edge e, its cnt[e] = max(#gt relations, 1) ranks already sorted and position-adjusted as the reference
 * appends them (first cnt[e] slots, rest 0).  Flattening the used slots in edge order gives exactly
 * the reference's result arrays.  Triplet ranks are counted along a staircase over each node's sorted class probabilities
 * (csrc/eval_ranks.hip): `scratch` holds vlsat_eval_ranks_scratch_floats(n_nodes, n_obj_class, topk_triplet) floats of device
 * memory the call may overwrite (the topk largest probabilities per node; required when n_edges > 0). */
int vlsat_eval_ranks(const float* obj_logits, const float* obj_probs, const float* rel_probs,
                     const int64_t* gt_class, const int64_t* gt_rel, const int64_t* edges,
                     int32_t n_nodes, int32_t n_edges, int32_t n_obj_class, int32_t n_rel_class,
                     int32_t topk_obj, int32_t topk_rel, int32_t topk_triplet, float threshold,
                     int32_t* obj_rank, int32_t* rel_rank, int32_t* tri_rank, int32_t* cnt, float* scratch, void* stream);
int64_t vlsat_eval_ranks_scratch_floats(int32_t n_nodes, int32_t n_obj_class, int32_t topk_triplet);

/* Rank arrays of one batch (the outputs of two vlsat_eval_ranks calls, 3D and 2D; cnt is the same for both: it depends on
 * the labels only) -> the ADDITIVE counts MMGNet.validation's summaries are functions of (reference src/model/model.py:
 * 214-242,267-282,364-388; get_mean_recall eva_utils_acc.py:224-237): counts[i] += ... for the 1 + R + 2 (11 + 6 R) = 361
 * fields of cvpr2023-vlsat_amd/evaluate.py fields() (scenes | cm_ge{1..R} | per branch: obj_n, obj_hit@{1,5,10}, rel_n,
 * rel_hit@{1,3,5}, tri_n, tri_hit@{50,100}, per predicate class n, tri@{50,100}, rel@{1,3,5}); obj_topk of the cls_matrix is the
 * 3D object rank for both branches (SGFN_MMG/model.py:469-470).  counts: device uint64, zeroed by the caller before the first
 * scene; integer atomics only, so scenes may be accumulated from several streams at once and the sums are exact.  This is
 * what lets a one-scene-per-call evaluation loop (validation()'s batch_size = 1, src/model/model.py:185) run without a host
 * round trip per scene.  All device pointers; asynchronous. */
int vlsat_eval_counts(const int32_t* obj_rank_3d, const int32_t* obj_rank_2d, const int32_t* rel_rank_3d, const int32_t* rel_rank_2d,
                      const int32_t* tri_rank_3d, const int32_t* tri_rank_2d, const int32_t* cnt, const int64_t* gt_class,
                      const int64_t* gt_rel, const int64_t* edges, int32_t n_nodes, int32_t n_edges, int32_t n_rel_class,
                      int32_t n_scenes, uint64_t* counts, void* stream);

/* One scene (or batch) of an evaluation loop in ONE call: vlsat_forward + softmax of the object logits + vlsat_eval_ranks for both
 * branches + vlsat_eval_counts, enqueued back to back on `stream` with every intermediate in the plan's own scratch -- what
 * Mmgnet.process_val (reference src/model/SGFN_MMG/model.py:458-480) computes per scene, reduced to what MMGNet.validation keeps of
 * it (src/model/model.py:201-242).  Top-k bounds 11 / 6 / 101 and threshold 0.5 are process_val's.  gt_rel is the multi-hot
 * [E, R] int64 target, edges_e2 the [E, 2] int64 (from, to) list in the plan's edge order, counts as for vlsat_eval_counts.
 * MODEL.multi_rel_outputs only.  All device pointers; asynchronous; one library call per scene on the host side. */
int vlsat_process_val_counts(vlsat_handle h, vlsat_plan plan, const float* obj_points, const float* obj_2d_feats,
                             const float* descriptor, const int64_t* gt_class, const int64_t* gt_rel, const int64_t* edges_e2,
                             int32_t n_scenes, uint64_t* counts, void* stream);

/* The additive fp64 metrics vector of one rank's batch -- what the path's one all-reduce carries when no labels are at hand
 * (bench.py; the label-based counts of validation(), reference src/model/model.py:214-242, come from vlsat_eval_ranks):
 * out9 = {n_scenes, n_nodes, n_edges, sum obj3d, sum obj2d, sum rel3d, sum rel2d, #nodes whose 3D and 2D top-1 class agree,
 * #edges whose 3D and 2D top-1 relation agree}.  All device pointers; the outputs are the dense [n_nodes, n_obj_class] /
 * [n_edges, n_rel_class] tensors vlsat_forward wrote; scratch holds 256 * 6 doubles.  Two launches on `stream`, fixed
 * summation order (no atomics): bit-reproducible. */
int vlsat_scene_checksums(const float* obj3d, const float* obj2d, int64_t n_nodes, int32_t n_obj_class, const float* rel3d,
                          const float* rel2d, int64_t n_edges, int32_t n_rel_class, int32_t n_scenes, double* out9, double* scratch,
                          void* stream);

/* -------- debug / test hooks (used by tests/ to localise a parity failure) -------------------- */
/* Stop vlsat_forward after stage `stage` (-1 = run everything).  Stages: 1 object encoder,
 * 2 node embedding, 3 edge embedding, 4 adapter, 5 distance bias, then for layer l:
 * 10+10l self-attention, +1 cross-attention, +2 gcn_3ds, +3 gcn_2ds, +4 edge cross-attention. */
int vlsat_debug_stop_after(vlsat_handle h, int32_t stage);
/* Device pointer and shape of a named workspace buffer of a plan ("X3","X2","E3","E2","F","G",
 * "AGG3","AGG2","H1","KP","NP","Hbig","bias","On","Oe","Qe","KVe").  "G" (gated values) and "AGG3"/"AGG2" keep
 * their 256 channels head-major (h*32 + m); the reference's order is m*8 + h. */
int vlsat_debug_buffer(vlsat_plan p, const char* name, void** ptr, int64_t* rows, int32_t* cols, int32_t* ld);
/* Synchronous strided device-to-device copy of that buffer into dst (row pitch dst_ld floats). */
int vlsat_debug_read(vlsat_plan p, const char* name, float* dst, int64_t dst_ld);

/* DVFS probe: while buf (device, >= 4*512 int64, zeroed) is non-NULL every block of the persistent GEMM
 * writes {shader cycles (s_memtime), 100 MHz wall ticks (s_memrealtime), tiles done, 1} when it exits;
 * cycles / (ticks / 1e8) is the shader clock the kernel actually ran at. NULL disables. */
int vlsat_debug_gemm_clock_probe(int64_t* buf);

/* Switches of one handle; defaults are the measured-best settings and none changes results beyond fp32 summation order / the
 * precision mode's rounding.  The RELEASE library accepts:
 *   "dual_stream" 0|1|2   extra streams for the 2D chains (1: launch-bound plans only, 2 = default: every plan; such a plan owns a second
 *                         scratch set, +11.3 KB per edge, and falls back to one stream past 48 GiB of workspace) -- plans created afterwards;
 *   "sched" -1|0|1        schedule of a multi-stream plan: 1 = dependency-exact, three lanes (3D chain / 2D edge chain / 2D node chain)
 *                         coupled by one event per data-flow edge; 0 = fork / join, lanes meet twice per layer; -1 (default) = exact in
 *                         the bf16 modes on plans of more than 16 384 edges, fork / join otherwise.  Bit-identical results;
 *   "flash_split" 0|1     split-key edge attention for plans that cannot fill the chip (plans created afterwards);
 *   "prof_dual" 0|1       per-class profiling keeps the multi-stream execution (1, default) or serialises on the launch stream;
 *   "pair_twins" 0|1, "pair_max_edges" n   plans of at most n edges (default 4096: one scene per call) run the 3D / 2D twin stages --
 *                         relation encoders, gcn_3ds | gcn_2ds, both head pairs -- as launches of two problems each, the edge cross-attention
 *                         of layer l on the second stream under the node attentions of layer l + 1 (round 6).  Bit-identical results;
 *   "gemm_p8", "gemm_dma", "gemm_splitk" 0|1          GEMM kernel selection (0: the older kernels);
 *   "gemm_splitk_max_tiles" n   the split-K kernel takes launches of at most n 64 x 64 output tiles (default 64; 0: half the resident slots);
 *   "gemm_p8_part_min" n   8-phase GEMM: remainder tiles (behind full rounds of 256 x 256 tiles) from which the launch takes them along as one
 *                         more, balanced round instead of leaving them to a tail launch; 0 = default by precision: 12 single-rounding bf16,
 *                         24 split-bf16, 5/8 of a round exact fp32 (profiles/r06_probes/ab_p8_part_min_*.txt);
 *   "gemm_k_rot" -1|0..7  8-phase GEMM: column tile tn of a row panel walks its K-tiles starting at tn * r (the blocks that share an A
 *                         panel ask L2 for its lines out of step); -1 = default: 1 for half-row bf16 launches, 0 otherwise.  Rotates an
 *                         fp32 summation order: inside every mode's tolerance (tests/test_hip_round6.py), not bit-identical;
 *   "gather_f16" -1|0|1   the node-side tables [P_i | P_j] that nn_edge.0 adds per edge (reference network_MMG.py:59-60,92: cat[x_i, e, x_j] as
 *                         three partial products) as fp16 half rows instead of fp32: half the gathered bytes, 2^-12 relative rounding of two
 *                         summands in front of a ReLU whose output is rounded to bf16 anyway; -1 = default: on in the single-rounding modes
 *                         (bf16_mixed, bf16), off in the split-bf16 ones, never in fp32;
 *   "outproj_f16" -1|0|1  the out-projection of a single-rounded edge attention (bf16_mixed, bf16, bf16x3_attn1) hands its rows to the
 *                         LayerNorm as fp16 half rows instead of fp32 (-1 = default: on);
 *   "split_fmt", "flash_bf16", "flash_tr", "pointnet_bf16", "gate_bf16", "ln_resid" 0|1   bf16 modes: tensor formats and kernels (0: the
 *                         fp32 forms) -- every one parity-tested both ways (tests/test_hip_forward.py);
 *   "flash_bq_big" 0|1    half-row edge attention, plans whose scenes all have >= 4096 edges: 256 queries per block (default) or 128;
 *   "flash_bq_big_min" n  ... that bound (plans created afterwards).  At the bench batch (1560 edges per scene) 256-query tiles are 1 %
 *                         slower per step than 128-query ones (profiles/r06_probes/ab_flash_bq_big_min.txt): the default stays 4096;
 *   "flash_qg" 0|1|2      half-row edge attention at head dim 64, no key split: 64 queries per wave (two 32-query groups share every K / V
 *                         fragment; 2 = group 0's P.V product in front of group 1's softmax).  Bit-identical to 0; 4-7 % slower per step
 *                         at BASELINE configs[2] and configs[4] (profiles/r06_probes/ab_flash_qg.txt): default 0;
 *   "flash_pv_terms" 3|2  split-bf16 edge attention: MFMAs per P.V product;  "gate_fuse_agg" 0|1|2: max aggregation inside the gate
 *                         kernel (never / bf16 modes / fp32 too);  "gate_row_map" 0|1, "gate_heads_mfma" 0|1|2: gate kernel variants.
 * Lab switches ("gate_grid", "gate_heads_bf16", "flash_heads_bf16", "node_attn_split", "half_fmt", "flash_dma", "flash_ablate" -- the
 * last one produces GARBAGE results for timing) exist in the experiments build only (`build.py --experiments`, -DVLSAT_EXPERIMENTS ->
 * tools/bin/libvlsat_hip_exp.so); the release library answers them with VLSAT_EINVAL. */
int vlsat_debug_option(vlsat_handle h, const char* name, int32_t value);

/* ---- the one collective of the path: sum of a short fp64 metrics vector over ranks, on RCCL ---------------
 * Scenes are sharded over ranks with no data-path exchange; a sharded evaluation ends with one all-reduce of additive
 * counts (what the reference's validation() computes its percentages from, src/model/model.py:214-242).  RCCL is
 * resolved at run time (the copy already in the process, e.g. PyTorch's, else librccl.so.1): no link-time dependency.
 * Bootstrap: rank 0 calls vlsat_comm_unique_id (128 bytes), the host shares them with the other ranks (file, pipe,
 * any launcher), every rank calls vlsat_comm_init with its GPU current. */
int vlsat_comm_unique_id(void* out128);
int vlsat_comm_init(const void* id128, int32_t n_ranks, int32_t rank, void** comm);
/* buf: device fp64[n], summed over ranks in place; asynchronous on `stream`. */
int vlsat_metrics_allreduce(void* comm, double* buf, int32_t n, void* stream);
void vlsat_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* VLSAT_H */
