// Forward orchestration of libvlsat_hip.so: Mmgnet.forward (reference src/model/SGFN_MMG/model.py:288-335) as a
// sequence of launches of the hand-written kernels of this directory on the caller's stream (plus, for small plans,
// a second stream for the 2D twin stages), and the per-kernel-class HIP-event profiling bench.py reads.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "engine.h"

using namespace vlsat;

static Scratch scratch_of(const vlsat_plan_s* p, int branch) {
    if (branch == 1 && p->dual)
        return {p->NP2, p->Hbig2, p->KP2, p->G2, p->T768b, p->Hbig2, p->Hbig2 + (size_t)std::max<int64_t>(p->E, 1) * 512, p->rs2, p->H2b};
    return {p->NP, p->Hbig, p->KP, p->G, p->T768, p->R1, p->R2, p->rs, p->H2};
}

namespace {

// bf16 modes keep the edge-row tensors that only feed matrix kernels (E3, E2, Hbig, the relation-head hiddens, Q / K|V / O
// of the edge attention) in the split-pair format (common.h pack_split): producers pack in their epilogues, consumers
// separate hi / lo with two v_perm_b32 instead of re-splitting fp32 on every fragment read.  Not with
// MODEL.feature_transform (its encoder chain keeps fp32 rows) and not when a debug option disabled one of the kernels
// that understand the format.
// Returns the format code: 0 fp32, 1 split-pair words (split-bf16 mode), 2 half rows = plain bf16 at half the HBM
// traffic (single-rounding modes, whose kernels never read a low part).
static int split_fmt(const vlsat_ctx* h) {
    // (head geometries other than 8 x 64: the chain tensors keep the format; Q / K|V / O of the edge attention and the gate's
    //  kproj stay fp32 there, because those kernels are the fp32 ones -- see SA in the edge attention below)
    const bool on = h->prec_edge != 0 && h->split_fmt && h->flash_bf16 && h->flash_tr && !h->gemm_no_dma && !h->d.feature_transform;
    if (!on) return 0;
    if (h->prec_edge == 3) return 1;
    return h->gate_bf16 && h->half_fmt ? 2 : 1;
}

// ---- profiling helpers ----
hipEvent_t next_event(vlsat_ctx* h) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}
// Per-class timing with as few events as possible: on each stream an event is recorded only where the kernel CLASS changes
// (a run of consecutive launches of one class is one interval), plus one where the stream's work ends (join / end of the
// forward).  With the 2D twin stages on the side stream the intervals of the two streams overlap; vlsat_profile_read puts
// them on one timeline (a base event per batch of records) and charges a class the length of the UNION of its intervals.
static inline int prof_lane(const vlsat_ctx* h, hipStream_t s) { return (h->side && s == h->side) ? 1 : (h->side2 && s == h->side2) ? 2 : 0; }
struct Scope {
    vlsat_ctx* h;
    hipStream_t s;
    int cls, w;
    double flops;
    long k0 = 0;
    Scope(vlsat_ctx* h_, hipStream_t s_, int cls_, double fl) : h(h_), s(s_), cls(cls_), w(prof_lane(h_, s_)), flops(fl) {
        k0 = h->gemm_launches;
        if (!h->prof) return;
        if (h->open_ok[w] && h->open[w].cls == cls) return;        // same class: the open interval continues
        if (!h->prof_base_set) {                                   // first record of a batch: the time origin
            if (!h->prof_base) hipEventCreate(&h->prof_base);
            hipEventRecord(h->prof_base, s);
            h->prof_base_set = true;
        }
        hipEvent_t e = next_event(h);
        hipEventRecord(e, s);
        if (h->open_ok[w]) { h->open[w].b = e; h->recs.push_back(h->open[w]); }
        h->open[w] = {cls, e, e, 0.0, 0};
        h->open_ok[w] = true;
    }
    ~Scope() {
        if (!h->prof) return;
        h->open[w].flops += flops;
        h->open[w].kernels += cls == PC_GEMM ? h->gemm_launches - k0 : 1;
    }
};
// end of a stream's work (join, end of a forward, a debug-stopped one): close its open interval
static void profile_close(vlsat_ctx* h, hipStream_t s) {
    const int w = prof_lane(h, s);
    if (!h->prof || !h->open_ok[w]) return;
    hipEvent_t e = next_event(h);
    hipEventRecord(e, s);
    h->open[w].b = e;
    h->recs.push_back(h->open[w]);
    h->open_ok[w] = false;
}

// Every nn.Linear of the path.  Operand precision follows the handle's mode: node-row launches (M == number of nodes
// of the running plan) take prec_node, everything else (edge rows, point rows) prec_edge; the bf16 planes of the
// weights were made when the mode was set (engine_weights.hip), so nothing is allocated or converted here.
// prec_override >= 0: this launch's operand precision whatever the row class says (the edge attention's projections in mode 4)
// fills in what a launch of this handle needs besides the problem itself: operand precision and bf16 weight planes, debug switches,
// the split-K workspace of lane `ws_lane`
static int gemm_resolve(vlsat_ctx* h, int ws_lane, GemmArgs& a, int prec_override) {
    const bool edge_fmt = a.a_split || a.c_split || a.r_split;            // tensors in an edge format: an edge-row launch whatever M is
    const int prec = prec_override >= 0 ? prec_override : (a.M == h->cur_N && !edge_fmt) ? h->prec_node : h->prec_edge;
    if (prec) {
        auto it = h->split.find(a.W);
        if (it == h->split.end()) return fail(VLSAT_ESTATE, "gemm: weight has no bf16 planes (set the precision after loading weights)");
        a.prec = prec;
        a.Whi = it->second.first;
        a.Wlo = it->second.second;
    }
    if (h->half_f16) {                      // mode 5: the half-row format is fp16 (engine.h)
        if (a.r_split == 2) return fail(VLSAT_ESTATE, "gemm: a half-row residual is not built on fp16 (the LayerNorm adds it)");
        if (a.a_split == 2) {               // fp16 operands: fp16 weight plane, f16 MFMA
            if (prec != 1) return fail(VLSAT_ESTATE, "gemm: half-row operands need the single-rounding precision");
            auto f = h->f16w.find(a.W);
            if (f == h->f16w.end()) return fail(VLSAT_ESTATE, "gemm: weight has no fp16 plane");
            a.Whi = f->second;
            a.half_f16 = 1;
        }
        if (a.c_split == 2) { a.c_split = 0; a.c_f16_cols = a.N; }      // half-row outputs: fp16, whatever the operands were
    }
    a.no_dma = h->gemm_no_dma;
    a.no_p8 = h->gemm_no_p8;
    a.p8_part_min = h->gemm_p8_part_min;
    a.k_rot = h->gemm_k_rot >= 0 ? h->gemm_k_rot : (a.prec == 1 && a.a_split == 2) ? 1 : 0;      // (round 6: +1 % per bf16_mixed step, nothing in the other modes)
    a.launches = &h->gemm_launches;
    if (h->gemm_splitk) {
        a.sk_ws = h->sk_ws[ws_lane]; a.sk_ws_floats = SPLITK_WS_FLOATS;
        a.sk_counters = h->sk_cnt[ws_lane]; a.sk_n_counters = SPLITK_COUNTERS;
        a.sk_max_tiles = h->gemm_splitk_max_tiles;
    }
    return 0;
}
int gemm(vlsat_ctx* h, hipStream_t s, const GemmArgs& a0, int prec_override = -1) {
    GemmArgs a = a0;
    RUN(gemm_resolve(h, prof_lane(h, s), a, prec_override));   // (a split-K workspace per lane: launches of different lanes overlap)
    Scope sc(h, s, PC_GEMM, gemm_flops(a));
    return launch_gemm(a, s);
}
// The 3D / 2D twins of a stage (same shape, same flags, different tensors and weights) on the caller's stream: ONE launch when the
// problems are small enough for the single-round kernels (launch_gemm_pair), else one after the other -- the same kernels on the
// same data either way, so the results do not depend on which it was.  The second problem takes lane 2's split-K workspace (the
// paired schedule never runs the third lane).
int gemm2(vlsat_ctx* h, hipStream_t s, const GemmArgs& a0, const GemmArgs& b0) {
    GemmArgs a = a0, b = b0;
    RUN(gemm_resolve(h, prof_lane(h, s), a, -1));
    RUN(gemm_resolve(h, 2, b, -1));
    {
        Scope sc(h, s, PC_GEMM, gemm_flops(a) + gemm_flops(b));
        const int r = launch_gemm_pair(a, b, s);
        if (r <= 0) return r;
    }
    b.sk_ws = a.sk_ws; b.sk_counters = a.sk_counters;           // (one stream: the launches are ordered)
    { Scope sc(h, s, PC_GEMM, gemm_flops(a)); RUN(launch_gemm(a, s)); }
    Scope sc(h, s, PC_GEMM, gemm_flops(b));
    return launch_gemm(b, s);
}

GemmArgs G(const float* A, int lda, const float* W, int K, float* C, int ldc, int M, int N, const float* bias,
           int act = ACT_NONE) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act;
    return g;
}

// One MultiHeadAttention block on node rows (reference transformer/attention.py:105-126), in place on xq.
// self: q = k = v = xq, one fused [3D, D] projection into qbuf = QKVn [N, 3D].  Cross (q = X2, k = v = X3): the key | value
// projection `kv` [N, 2D] was made by kv_project() on the lane that owns X3; only the query is projected here, into qbuf [N, D].
int attn_block(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const AttnW& w, float* xq, bool self, float* qbuf, const float* kv,
               float* obuf) {
    const int N = (int)p->N, D = h->D, LDX = ldx_of(h);
    const float *K, *V;
    int ldq, ldkv;
    if (self) {
        RUN(gemm(h, s, G(xq, LDX, w.wqkv, D, qbuf, 3 * D, N, 3 * D, w.bqkv)));
        ldq = ldkv = 3 * D; K = qbuf + D; V = qbuf + 2 * D;
    } else {
        RUN(gemm(h, s, G(xq, LDX, w.wq, D, qbuf, D, N, D, w.bq)));
        ldq = D; ldkv = 2 * D; K = kv; V = kv + D;
    }
    {
        Scope sc(h, s, PC_NODE_ATTN, 0);
        RUN(launch_node_attn(qbuf, ldq, K, ldkv, V, ldkv, obuf, D, p->bias,
                             p->d_scene_ptr, p->d_bias_ptr, p->S, p->max_n, h->H, D / h->H, 1.0f, s, h->node_attn_split));
    }
    GemmArgs o = G(obuf, D, w.wo, D, xq, LDX, N, D, w.bo);
    o.resid = xq; o.ldr = LDX;
    RUN(gemm(h, s, o));
    {
        Scope sc(h, s, PC_LAYERNORM, 0);
        RUN(launch_layernorm(xq, LDX, N, D, w.lng, w.lnb, 0, s));
    }
    return 0;
}
// key | value projection of a node cross-attention: kv [N, 2D] = xkv . Wkv^T + b
int kv_project(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const AttnW& w, const float* xkv, float* kv) {
    const int N = (int)p->N, D = h->D;
    return gemm(h, s, G(xkv, ldx_of(h), w.wkv, D, kv, 2 * D, N, 2 * D, w.bkv));
}

// node side of a GraphEdgeAttenNetwork block: NP [N, 6D + A] = [P_i | P_j | Gq | value]
int gcn_node_project(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const GcnW& w, const float* x, const Scratch& sc) {
    const int N = (int)p->N, D = h->D, NPC = npc_of(h);
    GemmArgs np = G(x, ldx_of(h), w.wnode, D, sc.NP, NPC, N, NPC, w.bnode);
    if (gather_f16_on(h)) np.c_f16_cols = 4 * D;                 // [P_i | P_j] as fp16 half rows (what nn_edge.0 gathers per edge: half the bytes)
    return gemm(h, s, np);
}

// node_done: sc.NP has already been filled by gcn_node_project (on another lane; the caller has ordered this lane behind it)
int gcn_block(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const GcnW& w, float* x, float* e, int e_relu_pending,
              int out_relu, const Scratch& sc, bool node_done = false) {
    const int N = (int)p->N, E = (int)p->E, D = h->D, A = h->A, LDX = ldx_of(h), NPC = npc_of(h);
    if (!node_done) RUN(gcn_node_project(h, p, s, w, x, sc));
    const int S = split_fmt(h);
    const bool gate16 = h->prec_edge && h->gate_bf16 && default_heads(h);
    const bool gate16h = h->prec_edge && h->gate_bf16 && (!default_heads(h) || h->gate_heads_mfma == 2) && h->gate_heads_mfma && h->gate_heads_bf16 &&
                         edge_gate_bf16_heads_supports(D / h->H, A / h->H, h->prec_edge == 3 ? 3 : 1);   // other head geometries, bf16 kernel
    GemmArgs e1 = G(e, D, w.we1, D, sc.Hbig, 2 * D, E, 2 * D, nullptr, ACT_RELU);
    e1.relu_a = e_relu_pending;
    e1.a_split = S; e1.c_split = S;
    e1.g0 = sc.NP; e1.gi0 = p->d_src; e1.ldg0 = NPC;
    e1.g_f16 = gather_f16_on(h);
    e1.g1 = sc.NP + (e1.g_f16 ? D : 2 * D); e1.gi1 = p->d_dst; e1.ldg1 = NPC;       // (fp16 half rows: P_j starts at byte 2 * 2D of the row)
    RUN(gemm(h, s, e1));
    if (h->d.use_gcn_edge) {              // proj_edge feeds only the gate MLP (reference network_MMG.py:98-102)
        GemmArgs kp = G(e, D, w.wpe, D, sc.KP, D, E, D, w.bpe);
        kp.relu_a = e_relu_pending;
        kp.a_split = S;
        kp.c_split = (gate16 || gate16h) ? S : 0;      // (the fp32 gate kernels read plain fp32)
        RUN(gemm(h, s, kp));
    }
    bool fused_agg = false;
    GemmArgs e2 = G(sc.Hbig, 2 * D, w.we2, 2 * D, e, D, E, D, w.be2);       // e <- nn_edge output (pre-activation)
    e2.a_split = S; e2.c_split = S;
    RUN(gemm(h, s, e2));
    {
        GateArgs g{};
        g.kproj = sc.KP; g.node = sc.NP; g.ld_node = NPC; g.gq_off = 4 * D; g.v_off = 6 * D;    // Gq spans H * 2 d_k = 2 D columns
        g.src = p->d_src; g.dst = p->d_dst; g.w0k = w.w0k; g.w3 = w.w3; g.b3 = w.b3; g.gated = sc.G;
        g.prob = p->prob; g.n_edges = E; g.use_edge = h->d.use_gcn_edge; g.grid_cap = h->gate_grid; g.row_map = h->gate_row_map;
        const double dk = D / h->H, dox = A / h->H;
        // the gate at 8 x (64, 32) in any precision, max aggregation, no debug tap: the aggregation happens inside the gate kernel (no [E, 256]
        // tensor of gated messages, no aggregate launch); the start values go in first
        const bool shipped_kernel = !gate16h && default_heads(h) && !(h->gate_heads_mfma == 2 && !gate16);     // edge_gate.hip / edge_gate_bf16.hip
        // (fp32: measured neutral -- 2196-2201 vs 2195 scenes/s -- and it moves waiting time into the GEMM class of the two-stream
        //  profile, so the exact-fp32 mode keeps the separate aggregate launch unless "gate_fuse_agg" is 2)
        fused_agg = shipped_kernel && (h->gate_fuse_agg == 2 || (h->gate_fuse_agg == 1 && gate16)) && h->d.gcn_aggr == 0 && !g.prob && h->debug_stop < 0 && h->gate_row_map && E > 0;
        if (fused_agg) {
            Scope scope(h, s, PC_AGGREGATE, 0);
            RUN(launch_agg_init(p->d_rowptr, N, A, x + D, LDX, s));
            g.agg = x + D; g.ld_agg = LDX;
        }
        Scope scope(h, s, PC_GATE, (double)E * h->H * (2.0 * dk * 2 * dk + 2.0 * 2 * dk * dox));
        if (gate16h) {
            RUN(launch_edge_gate_bf16_heads(g, h->H, D / h->H, A / h->H, h->prec_edge == 3 ? 3 : 1, (S == 2 && h->half_f16) ? 3 : S, s));
        } else if (!default_heads(h) || (h->gate_heads_mfma == 2 && !gate16)) {     // (2: the shipped geometry on the template as well -- A/B)
            int r = h->gate_heads_mfma ? launch_edge_gate_heads(g, h->H, D / h->H, A / h->H, s) : 1;
            if (r < 0) return r;
            if (r) RUN(launch_edge_gate_generic(g, h->H, D / h->H, A / h->H, s));
        }
        else if (gate16) RUN(launch_edge_gate_bf16(g, h->prec_edge == 3 ? 3 : 1, (S == 2 && h->half_f16) ? 3 : S, s));
        else RUN(launch_edge_gate(g, s));
    }
    if (!fused_agg) {
        Scope scope(h, s, PC_AGGREGATE, 0);
        RUN(launch_aggregate(sc.G, A, p->d_rowptr, p->d_order, N, h->d.gcn_aggr, x, LDX, D, s));
    }
    RUN(gemm(h, s, G(x, LDX, w.wp0, D + A, sc.T768, D + A, N, D + A, w.bp0, ACT_RELU)));
    RUN(gemm(h, s, G(sc.T768, D + A, w.wp2, D + A, x, LDX, N, D, w.bp2, out_relu ? ACT_RELU : ACT_NONE)));
    return 0;
}

int rel_head(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const RelHeadW& w, const float* e, int relu_a, float* out,
             const Scratch& sc) {
    const int E = (int)p->E, D = h->D, R = h->d.n_rel_class;
    const int S = split_fmt(h);
    GemmArgs a = G(e, D, w.w1, D, sc.R1, 512, E, 512, w.b1, ACT_RELU);
    a.relu_a = relu_a;
    a.a_split = S; a.c_split = S;
    RUN(gemm(h, s, a));
    GemmArgs b = G(sc.R1, 512, w.w2, 512, sc.R2, 256, E, 256, w.b2, ACT_RELU);
    b.a_split = S; b.c_split = S;
    RUN(gemm(h, s, b));
    // multi_rel_outputs: sigmoid (PointNetRelClsMulti) or log_softmax over the R classes (PointNetRelCls)
    GemmArgs c = G(sc.R2, 256, w.w3, 256, out, R, E, R, w.b3, h->d.multi_rel_outputs ? ACT_SIGMOID : ACT_NONE);
    c.a_split = S;
    RUN(gemm(h, s, c));
    if (!h->d.multi_rel_outputs) {
        Scope scope(h, s, PC_MISC, 0);
        RUN(launch_softmax_rows(out, R, E, R, out, 1, s));
    }
    return 0;
}

int obj_head(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const float* x, const float* w, const float* b, float* out,
             const Scratch& sc) {
    const int N = (int)p->N, D = h->D, C = h->d.n_obj_class;
    {
        Scope scope(h, s, PC_MISC, 0);
        RUN(launch_row_invnorm(x, ldx_of(h), N, D, std::exp(h->d.obj_logit_scale), sc.rs, s));
    }
    GemmArgs a = G(x, ldx_of(h), w, D, out, C, N, C, b);
    a.rowscale = sc.rs;
    RUN(gemm(h, s, a));
    return 0;
}

// ---- the paired schedule of one-scene plans (round 6): gcn_3ds[l] and gcn_2ds[l], the two relation heads, the two object heads and
// the two relation encoders as launches of TWO problems each (gemm2, twin gate / aggregate launches).  Stage by stage the same
// kernels on the same operands as gcn_block / rel_head / obj_head above: bit-identical outputs.
// Shipped gate kernels only (default head geometry, no probability tap); the caller checks.
int gcn_block_pair(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const GcnW& w3, const GcnW& w2, float* x3, float* x2, float* e3, float* e2,
                   int e3_relu_pending, int out_relu, const Scratch& sc3, const Scratch& sc2) {
    const int N = (int)p->N, E = (int)p->E, D = h->D, A = h->A, LDX = ldx_of(h), NPC = npc_of(h);
    const int S = split_fmt(h);
    const bool gate16 = h->prec_edge && h->gate_bf16;
    {
        GemmArgs n3 = G(x3, LDX, w3.wnode, D, sc3.NP, NPC, N, NPC, w3.bnode), n2 = G(x2, LDX, w2.wnode, D, sc2.NP, NPC, N, NPC, w2.bnode);
        if (gather_f16_on(h)) n3.c_f16_cols = n2.c_f16_cols = 4 * D;
        RUN(gemm2(h, s, n3, n2));
    }
    auto e1_of = [&](const GcnW& w, float* e, int relu, const Scratch& sc) {
        GemmArgs e1 = G(e, D, w.we1, D, sc.Hbig, 2 * D, E, 2 * D, nullptr, ACT_RELU);
        e1.relu_a = relu;
        e1.a_split = S; e1.c_split = S;
        e1.g0 = sc.NP; e1.gi0 = p->d_src; e1.ldg0 = NPC;
        e1.g_f16 = gather_f16_on(h);
        e1.g1 = sc.NP + (e1.g_f16 ? D : 2 * D); e1.gi1 = p->d_dst; e1.ldg1 = NPC;
        return e1;
    };
    RUN(gemm2(h, s, e1_of(w3, e3, e3_relu_pending, sc3), e1_of(w2, e2, 0, sc2)));
    if (h->d.use_gcn_edge) {
        auto kp_of = [&](const GcnW& w, float* e, int relu, const Scratch& sc) {
            GemmArgs kp = G(e, D, w.wpe, D, sc.KP, D, E, D, w.bpe);
            kp.relu_a = relu;
            kp.a_split = S;
            kp.c_split = gate16 ? S : 0;
            return kp;
        };
        RUN(gemm2(h, s, kp_of(w3, e3, e3_relu_pending, sc3), kp_of(w2, e2, 0, sc2)));
    }
    auto e2_of = [&](const GcnW& w, float* e, const Scratch& sc) {
        GemmArgs g = G(sc.Hbig, 2 * D, w.we2, 2 * D, e, D, E, D, w.be2);
        g.a_split = S; g.c_split = S;
        return g;
    };
    RUN(gemm2(h, s, e2_of(w3, e3, sc3), e2_of(w2, e2, sc2)));
    auto gate_of = [&](const GcnW& w, const Scratch& sc) {
        GateArgs g{};
        g.kproj = sc.KP; g.node = sc.NP; g.ld_node = NPC; g.gq_off = 4 * D; g.v_off = 6 * D;
        g.src = p->d_src; g.dst = p->d_dst; g.w0k = w.w0k; g.w3 = w.w3; g.b3 = w.b3; g.gated = sc.G;
        g.prob = nullptr; g.n_edges = E; g.use_edge = h->d.use_gcn_edge; g.grid_cap = h->gate_grid; g.row_map = h->gate_row_map;
        return g;
    };
    GateArgs g3 = gate_of(w3, sc3), g2 = gate_of(w2, sc2);
    const double dk = D / h->H, dox = A / h->H;
    const bool fused_agg = (h->gate_fuse_agg == 2 || (h->gate_fuse_agg == 1 && gate16)) && h->d.gcn_aggr == 0 && h->gate_row_map && E > 0;
    if (fused_agg) {
        Scope scope(h, s, PC_AGGREGATE, 0);
        RUN(launch_agg_init(p->d_rowptr, N, A, x3 + D, LDX, s, x2 + D));
        g3.agg = x3 + D; g3.ld_agg = LDX;
        g2.agg = x2 + D; g2.ld_agg = LDX;
    }
    {
        Scope scope(h, s, PC_GATE, 2.0 * (double)E * h->H * (2.0 * dk * 2 * dk + 2.0 * 2 * dk * dox));
        if (gate16) RUN(launch_edge_gate_bf16(g3, h->prec_edge == 3 ? 3 : 1, (S == 2 && h->half_f16) ? 3 : S, s, &g2));
        else RUN(launch_edge_gate(g3, s, &g2));
    }
    if (!fused_agg) {
        Scope scope(h, s, PC_AGGREGATE, 0);
        RUN(launch_aggregate(sc3.G, A, p->d_rowptr, p->d_order, N, h->d.gcn_aggr, x3, LDX, D, s, sc2.G, x2));
    }
    RUN(gemm2(h, s, G(x3, LDX, w3.wp0, D + A, sc3.T768, D + A, N, D + A, w3.bp0, ACT_RELU), G(x2, LDX, w2.wp0, D + A, sc2.T768, D + A, N, D + A, w2.bp0, ACT_RELU)));
    RUN(gemm2(h, s, G(sc3.T768, D + A, w3.wp2, D + A, x3, LDX, N, D, w3.bp2, out_relu ? ACT_RELU : ACT_NONE),
              G(sc2.T768, D + A, w2.wp2, D + A, x2, LDX, N, D, w2.bp2, out_relu ? ACT_RELU : ACT_NONE)));
    return 0;
}

int rel_head_pair(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const float* e3, int relu3, float* out3, const float* e2, float* out2,
                  const Scratch& sc3, const Scratch& sc2) {
    const int E = (int)p->E, D = h->D, R = h->d.n_rel_class;
    const int S = split_fmt(h);
    auto fc1 = [&](const RelHeadW& w, const float* e, int relu, const Scratch& sc) {
        GemmArgs a = G(e, D, w.w1, D, sc.R1, 512, E, 512, w.b1, ACT_RELU);
        a.relu_a = relu; a.a_split = S; a.c_split = S;
        return a;
    };
    auto fc2 = [&](const RelHeadW& w, const Scratch& sc) {
        GemmArgs b = G(sc.R1, 512, w.w2, 512, sc.R2, 256, E, 256, w.b2, ACT_RELU);
        b.a_split = S; b.c_split = S;
        return b;
    };
    auto fc3 = [&](const RelHeadW& w, const Scratch& sc, float* out) {
        GemmArgs c = G(sc.R2, 256, w.w3, 256, out, R, E, R, w.b3, h->d.multi_rel_outputs ? ACT_SIGMOID : ACT_NONE);
        c.a_split = S;
        return c;
    };
    RUN(gemm2(h, s, fc1(h->rel3, e3, relu3, sc3), fc1(h->rel2, e2, 0, sc2)));
    RUN(gemm2(h, s, fc2(h->rel3, sc3), fc2(h->rel2, sc2)));
    RUN(gemm2(h, s, fc3(h->rel3, sc3, out3), fc3(h->rel2, sc2, out2)));
    if (!h->d.multi_rel_outputs) {
        Scope scope(h, s, PC_MISC, 0);
        RUN(launch_softmax_rows(out3, R, E, R, out3, 1, s));
        RUN(launch_softmax_rows(out2, R, E, R, out2, 1, s));
    }
    return 0;
}

int obj_head_pair(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, float* out3, float* out2, const Scratch& sc3, const Scratch& sc2) {
    const int N = (int)p->N, D = h->D, C = h->d.n_obj_class;
    {
        Scope scope(h, s, PC_MISC, 0);
        RUN(launch_row_invnorm(p->X3, ldx_of(h), N, D, std::exp(h->d.obj_logit_scale), sc3.rs, s, p->X2, sc2.rs));
    }
    GemmArgs a = G(p->X3, ldx_of(h), h->obj3_w, D, out3, C, N, C, h->obj3_b), b = G(p->X2, ldx_of(h), h->obj2_w, D, out2, C, N, C, h->obj2_b);
    a.rowscale = sc3.rs;
    b.rowscale = sc2.rs;
    return gemm2(h, s, a, b);
}

int stn_encoder(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const StnW& w, const float* h1, int ldh, size_t R, int P,
                const float* w2, const float* b2, const float* w3, const float* b3, int n_out, float** out_rows) {
    const size_t O = R / P;
    float* ws = p->stn_ws;
    float* a64 = ws;             ws += R * 64;
    float* a128 = ws;            ws += R * 128;
    float* a1024 = ws;           ws += R * 1024;
    float* h1t = ws;             ws += R * 64;
    float* g = ws;               ws += O * 1024;
    float* f1 = ws;              ws += O * 512;
    float* f2 = ws;              ws += O * 256;
    float* T = ws;               ws += O * 4096;
    if ((size_t)(ws - p->stn_ws) > p->stn_ws_floats) return fail(VLSAT_ESTATE, "feature_transform scratch too small");
    const int Ri = (int)R, Oi = (int)O;
    RUN(gemm(h, s, G(h1, ldh, w.c1, 64, a64, 64, Ri, 64, w.c1b, ACT_RELU)));
    RUN(gemm(h, s, G(a64, 64, w.c2, 64, a128, 128, Ri, 128, w.c2b, ACT_RELU)));
    RUN(gemm(h, s, G(a128, 128, w.c3, 128, a1024, 1024, Ri, 1024, w.c3b, ACT_RELU)));
    const float* gp = a1024;
    if (P > 1) {
        Scope sc(h, s, PC_MISC, 0);
        RUN(launch_rowmax(a1024, 1024, Oi, P, 1024, g, 1024, s));
        gp = g;
    }
    RUN(gemm(h, s, G(gp, 1024, w.f1, 1024, f1, 512, Oi, 512, w.f1b, ACT_RELU)));
    RUN(gemm(h, s, G(f1, 512, w.f2, 512, f2, 256, Oi, 256, w.f2b, ACT_RELU)));
    RUN(gemm(h, s, G(f2, 256, w.f3, 256, T, 4096, Oi, 4096, w.f3b)));
    {
        Scope sc(h, s, PC_MISC, 0);
        RUN(launch_apply_stn(h1, ldh, T, R, P, h1t, 64, s));
    }
    RUN(gemm(h, s, G(h1t, 64, w2, 64, a128, 128, Ri, 128, b2, ACT_RELU)));
    RUN(gemm(h, s, G(a128, 128, w3, 128, a1024, n_out, Ri, n_out, b3, ACT_RELU)));
    *out_rows = a1024;
    return 0;
}

}  // namespace

// ============================================================================================
extern "C" {

// -------------------------------------------------------------------------------------------
struct TrainOut { float *mimic3d, *mimic2d, *edge_dis; };

static int forward_body(vlsat_handle h, vlsat_plan p, const float* pts, const float* f2d, const float* desc,
                        float* obj3d, float* obj2d, float* rel3d, float* rel2d, const TrainOut* tr, void* stream);

// A forward that fails half way has already enqueued kernels on the plan's arena (possibly on the side stream too): every
// exit path joins the side stream into the caller's and records the plan's last-use event, so that vlsat_plan_destroy never
// recycles the arena behind a stale or never-recorded event (the contract of vlsat.h).
static int forward_impl(vlsat_handle h, vlsat_plan p, const float* pts, const float* f2d, const float* desc,
                        float* obj3d, float* obj2d, float* rel3d, float* rel2d, const TrainOut* tr, void* stream) {
    const int rc = forward_body(h, p, pts, f2d, desc, obj3d, obj2d, rel3d, rel2d, tr, stream);
    if (rc != 0 && h && p && p->h == h && p->used) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        const std::string msg = last_error_cstr();                    // (the calls below must not replace the real error)
        for (hipStream_t side : {h->side, h->side2}) {
            hipEvent_t e = nullptr;
            if (side && p->dual && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) {
                hipEventRecord(e, side);
                hipStreamWaitEvent(s, e, 0);
                hipEventDestroy(e);                                   // (destruction is deferred until the event completes)
            }
        }
        h->open_ok[0] = h->open_ok[1] = h->open_ok[2] = false;        // profiling intervals left open are dropped
        hipEventRecord(p->last_use, s);
        set_error(msg);
    }
    return rc;
}

static int forward_body(vlsat_handle h, vlsat_plan p, const float* pts, const float* f2d, const float* desc,
                        float* obj3d, float* obj2d, float* rel3d, float* rel2d, const TrainOut* tr, void* stream) {
    if (!h || !p || !pts || !desc || !obj3d) return fail(VLSAT_EINVAL, "vlsat_forward: null argument");
    if (p->h != h) return fail(VLSAT_EINVAL, "plan belongs to a different handle");
    // a weight reload is open (vlsat_load_weight after a finalize freed the device weights): nothing may launch on them
    if (!h->finalized) return fail(VLSAT_ESTATE, "vlsat_forward: weights are not finalised (a reload is in progress or failed)");
    // 3D-only mode: both 2D outputs NULL -> the 2D branch (adapter, cross-attention, gcn_2ds, edge
    // cross-attention, 2D heads) is skipped.  Exact: the 3D branch never reads 2D tensors.
    const bool do2d = obj2d != nullptr || rel2d != nullptr;
    if (do2d && (!obj2d || !f2d || (p->E > 0 && !rel2d)))
        return fail(VLSAT_EINVAL, "vlsat_forward: 2D branch needs obj_2d_feats and both 2D outputs (or neither for 3D-only)");
    if (p->E > 0 && !rel3d) return fail(VLSAT_EINVAL, "vlsat_forward: null relation output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = (int)p->N, E = (int)p->E, D = h->D, L = h->d.n_layers, LDX = ldx_of(h);
    const int stop = h->debug_stop;
    h->cur_N = N;
    if (p->upload_pending) {   // the plan's index tables travel on the handle's copy stream
        VLSAT_HIP_CHECK(hipStreamWaitEvent(s, p->uploaded, 0));
        if (hipEventQuery(p->uploaded) == hipSuccess) p->upload_pending = false;
    }
    p->used = true;
    profile_close(h, s);                   // an interval left open by a failed forward must not span foreign work
#define STAGE(id) do { if (stop == (id)) { profile_close(h, s); hipEventRecord(p->last_use, s); return 0; } } while (0)

    // Lanes.  A two-stream plan (p->dual: it owns a second scratch set) runs the forward on up to three streams that always end
    // joined into the caller's, so the caller only ever sees its own stream:
    //   s  the caller's stream: the 3D chain -- object encoder, self-attention, gcn_3ds, the key | value projections both
    //      cross-attentions read from 3D tensors, the 3D heads.  It never reads a 2D tensor (reference network_MMG.py:217-234,
    //      SURVEY 3.3), so it never waits for the other lanes except for a buffer to become free;
    //   t  the 2D edge chain: rel_encoder_2d, gcn_2ds (edge side + prop), edge cross-attention (query projection, attention,
    //      out-projection, LayerNorm), the 2D relation head;
    //   u  the 2D node chain: adapter, node cross-attention, the node-side projection of gcn_2ds, the 2D object head.
    // "sched" = 1 (default): dependency-exact -- every cross-lane edge of the data flow is ONE event (after() / wait()), so the
    // latency-bound node-row launches and GEMM tails of one lane execute under the non-persistent attention / gate kernels of
    // another.  "sched" = 0: the fork / join schedule of round 4 (u = t; the lanes meet twice per layer).  The launches, their
    // operands and therefore the results are the same in both; not while stopping at a debug stage or for the training outputs.
    const bool dual = p->dual && do2d && (!h->prof || h->prof_dual) && stop < 0 && !tr;
    // Paired schedule (round 6) for one-scene plans: a forward of a few thousand edges is ~114 launches of ~10 us that each leave most
    // of the chip idle, and a loop with several scenes in flight is bound by the SUM of their durations.  The 3D / 2D twin stages
    // (relation encoders, gcn_3ds | gcn_2ds, both head pairs) have the same shapes, so each pair becomes ONE launch of two problems
    // on the caller's stream; the second lane keeps the only chain without a twin, the edge cross-attention of layer l, which runs
    // under the node attentions of layer l + 1.  Same kernels on the same operands as the other schedules: bit-identical outputs.
    const bool pair = dual && h->pair_twins && p->E > 0 && p->E <= h->pair_max_edges && !h->d.feature_transform && default_heads(h) &&
                      !p->prob && !(h->gate_heads_mfma == 2 && !(h->prec_edge && h->gate_bf16)) && h->sched != 1;
    // (default: exact for the bf16 modes on plans that fill the chip by themselves.  One-scene plans keep two streams: with K replicas
    //  in flight on K host threads the third stream of every replica competes for the few hardware queues -- 551 vs 926-953 scenes/s at
    //  four in flight in bf16_mixed, profiles/r05_probes/val_loop_bf16_mixed_three_lanes.txt)
    const bool exact = dual && !pair && (h->sched < 0 ? (h->prec_edge != 0 && p->E > 16384) : h->sched != 0);
    hipStream_t t = s, u = s;
    size_t ev_i = 0;
    if (dual) {
        if (!h->side) VLSAT_HIP_CHECK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        // (the third lane exists only on handles that use it: HIP spreads streams over a few hardware queues in creation order, and
        //  an idle extra stream per replica cost the one-scene-per-call loop its scaling with scenes in flight -- 454 vs 570
        //  scenes/s at four in flight, profiles/r05_probes/val_loop_fp32.txt)
        if (exact && !h->side2) VLSAT_HIP_CHECK(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
        t = h->side;
        u = exact ? h->side2 : h->side;
    }
    auto next_ev = [&](hipEvent_t* e) -> int {
        if (ev_i == h->sync_ev.size()) {
            hipEvent_t n;
            VLSAT_HIP_CHECK(hipEventCreateWithFlags(&n, hipEventDisableTiming));
            h->sync_ev.push_back(n);
        }
        *e = h->sync_ev[ev_i++];
        return 0;
    };
    // after(x, &e): e stands for everything enqueued on lane x so far; wait(y, e): lane y continues behind it.  While the
    // per-class profile is on, a lane's open interval ends where it starts to wait (the wait is nobody's kernel time).
    auto after = [&](hipStream_t x, hipEvent_t* e) -> int {
        *e = nullptr;
        if (!dual) return 0;
        RUN(next_ev(e));
        VLSAT_HIP_CHECK(hipEventRecord(*e, x));
        return 0;
    };
    auto wait = [&](hipStream_t y, hipEvent_t e) -> int {
        if (!dual || !e) return 0;
        profile_close(h, y);
        VLSAT_HIP_CHECK(hipStreamWaitEvent(y, e, 0));
        return 0;
    };
    auto order = [&](hipStream_t first, hipStream_t then) -> int {       // `then` continues after `first`'s work so far
        if (!dual || first == then) return 0;
        hipEvent_t e;
        RUN(after(first, &e));
        return wait(then, e);
    };
    auto fork = [&]() { return (exact || pair) ? 0 : order(s, t); };     // (round-4 schedule only)
    auto join = [&]() { return (exact || pair) ? 0 : order(t, s); };
    const Scratch sc3 = scratch_of(p, 0), sc2 = scratch_of(p, dual ? 1 : 0);
    const int S = split_fmt(h);            // edge tensors in the split-pair format (bf16 modes)
    if (exact) RUN(order(s, u));           // u starts where the forward starts (behind whatever the caller enqueued before it)
    if (pair) RUN(order(s, t));            // (paired schedule: so does t -- the adapter, then the edge cross-attention of every layer)

    const bool ft = h->d.feature_transform != 0;
    if (!ft) {   // a-2 object encoder
        Scope sc(h, s, PC_POINTNET, 213376.0 * N * p->P);
        if (h->prec_edge && h->pointnet_bf16) {     // point rows count as edge-class work: the bf16 matrix cores
            auto w2 = h->split.find(h->pn_w2), w3 = h->split.find(h->pn_w3);
            if (w2 == h->split.end() || w3 == h->split.end()) return fail(VLSAT_ESTATE, "pointnet: weights have no bf16 planes");
            const uint16_t *w2h = w2->second.first, *w3h = w3->second.first;
            int terms = h->prec_edge == 3 ? 3 : 1;
            if (h->half_f16) {                       // mode 5: single-rounded fp16 operands (fp16 planes of conv2 / conv3)
                auto f2 = h->f16w.find(h->pn_w2), f3 = h->f16w.find(h->pn_w3);
                if (f2 == h->f16w.end() || f3 == h->f16w.end()) return fail(VLSAT_ESTATE, "pointnet: weights have no fp16 planes");
                w2h = f2->second; w3h = f3->second; terms = 2;
            }
            RUN(launch_pointnet_bf16(pts, N, p->P, h->d.dim_point, h->pn_w1, h->pn_b1, w2h, w2->second.second, h->pn_b2,
                                     w3h, w3->second.second, h->pn_b3, h->C_pt, terms, p->F, s));
        } else {
            RUN(launch_pointnet(pts, N, p->P, h->d.dim_point, h->pn_w1, h->pn_b1, h->pn_w2, h->pn_b2, h->pn_w3, h->pn_b3, h->C_pt, p->F, s));
        }
    } else {     // a-2 with the STNkd feature transform: conv1 as point rows, then GEMMs + a max over each object's rows
        float* rows = p->stn_ws + p->stn_ws_floats - (size_t)N * p->P * 64;       // h1 rows live at the end of the scratch
        {
            Scope sc(h, s, PC_MISC, 0);
            RUN(launch_pts_conv1_rows(pts, N, p->P, h->d.dim_point, h->pn_w1, h->pn_b1, rows, s));
        }
        float* out = nullptr;
        RUN(stn_encoder(h, p, s, h->stn_obj, rows, 64, (size_t)N * p->P, p->P, h->pn_w2, h->pn_b2, h->pn_w3, h->pn_b3, h->C_pt, &out));
        Scope sc(h, s, PC_MISC, 0);
        RUN(launch_rowmax(out, h->C_pt, N, p->P, h->C_pt, p->F, 768, s));
    }
    if (tr && tr->mimic3d)                 // obj_feature[..., :512] (reference SGFN_MMG/model.py:291-292)
        RUN(launch_copy_rows(tr->mimic3d, 512, p->F, 768, 512, (size_t)N, s));
    STAGE(1);
    // a-3 mlp_3d (+BN folded) + spatial tail -> X3[:, 0:512]
    RUN(gemm(h, s, G(p->F, 768, h->mlp_w, 768, p->X3, LDX, N, D - 8, h->mlp_b, ACT_RELU)));
    {
        Scope sc(h, s, PC_MISC, 0);
        RUN(launch_desc_tail(desc, N, p->X3, LDX, D - 8, s));
    }
    STAGE(2);
    // a-4/a-5 edge descriptor + relation encoders
    {
        Scope sc(h, s, PC_MISC, 0);
        RUN(launch_edge_embed(desc, p->d_src, p->d_dst, E, h->re_w1cat, h->re_b1cat, p->H1, s));
    }
    hipEvent_t h1_ready = nullptr;
    if (!pair) {
        RUN(after(s, &h1_ready));
        RUN(wait(t, h1_ready));                                             // t: 2D relation encoder (old schedule: + adapter)
    }
    if (pair) {                                                             // conv2 / conv3 of both relation encoders, two problems per launch
        GemmArgs c2a = G(p->H1, 128, h->re3_w2, 64, sc3.H2, 128, E, 128, h->re3_b2, ACT_RELU), c2b = G(p->H1 + 64, 128, h->re2_w2, 64, sc2.H2, 128, E, 128, h->re2_b2, ACT_RELU);
        c2a.c_split = S; c2b.c_split = S;
        RUN(gemm2(h, s, c2a, c2b));
        GemmArgs c3a = G(sc3.H2, 128, h->re3_w3, 128, p->E3, D, E, D, h->re3_b3, ACT_RELU), c3b = G(sc2.H2, 128, h->re2_w3, 128, p->E2, D, E, D, h->re2_b3, ACT_RELU);
        c3a.a_split = S; c3a.c_split = S; c3b.a_split = S; c3b.c_split = S;
        RUN(gemm2(h, s, c3a, c3b));
    } else if (!ft) {
        for (int br = 0; br < (do2d ? 2 : 1); ++br) {          // conv2 / conv3 of rel_encoder_3d (on s) and rel_encoder_2d (on t)
            const Scratch& sc = br ? sc2 : sc3;
            GemmArgs c2 = G(p->H1 + 64 * br, 128, br ? h->re2_w2 : h->re3_w2, 64, sc.H2, 128, E, 128, br ? h->re2_b2 : h->re3_b2, ACT_RELU);
            c2.c_split = S;
            RUN(gemm(h, br ? t : s, c2));
            GemmArgs c3 = G(sc.H2, 128, br ? h->re2_w3 : h->re3_w3, 128, br ? p->E2 : p->E3, D, E, D, br ? h->re2_b3 : h->re3_b3, ACT_RELU);
            c3.a_split = S; c3.c_split = S;
            RUN(gemm(h, br ? t : s, c3));
        }
    } else if (E > 0) {   // P = 1: one row per edge; both encoders share the scratch, so they run one after the other on s
        for (int br = 0; br < (do2d ? 2 : 1); ++br) {
            float* out = nullptr;
            RUN(stn_encoder(h, p, s, br ? h->stn_re2 : h->stn_re3, p->H1 + 64 * br, 128, (size_t)E, 1, br ? h->re2_w2 : h->re3_w2,
                            br ? h->re2_b2 : h->re3_b2, br ? h->re2_w3 : h->re3_w3, br ? h->re2_b3 : h->re3_b3, D, &out));
            RUN(launch_copy_rows(br ? p->E2 : p->E3, (size_t)D, out, (size_t)D, D, (size_t)E, s));
        }
        if (dual) { hipEvent_t e; RUN(after(s, &e)); RUN(wait(t, e)); }    // (E2 was written on s)
    }
    STAGE(3);
    // a-6 adapter -> X2[:, 0:512]   (lane u: it depends on the inputs only -- u was forked where the forward began)
    hipEvent_t x2_ready = nullptr;        // X2[:, 0:512] is complete (adapter, then prop of gcn_2ds[l]) -> the node cross-attention may run
    if (do2d) {
        RUN(gemm(h, u, G(f2d, D, h->ad_w1, D, p->T256, 256, N, 256, h->ad_b1, ACT_RELU)));
        GemmArgs a = G(p->T256, 256, h->ad_w2h, 256, p->X2, LDX, N, D, h->ad_b2h);
        a.resid = f2d; a.ldr = D; a.resid_scale = 0.5f;
        RUN(gemm(h, u, a));
        if (tr && tr->mimic2d)             // the adapter's output before the MMG touches it (:312)
            RUN(launch_copy_rows(tr->mimic2d, 512, p->X2, (size_t)LDX, 512, (size_t)N, u));
        if (exact || pair) RUN(after(u, &x2_ready));
    }
    STAGE(4);
    {   // a-7 distance bias
        Scope sc(h, s, PC_MISC, 0);
        RUN(launch_dist_bias(desc, 11, p->d_scene_ptr, p->d_bias_ptr, p->S, p->max_n, h->H, h->db, p->bias, s));
    }
    STAGE(5);
    int e3_pending_relu = 0;
    hipEvent_t flash_done[2] = {nullptr, nullptr};            // the edge attention that read KVe slot i has completed
    hipEvent_t edge_done = nullptr;                           // (paired schedule) the edge cross-attention of the previous layer has written E2 and is done with E3
    for (int l = 0; l < L; ++l) {
        const int inter = (l < L - 1 || L == 1) ? 1 : 0;     // reference network_MMG.py:236
        const int base = 10 + 10 * l;
        RUN(attn_block(h, p, s, h->self_attn[l], p->X3, true, p->QKVn, nullptr, p->On));   // :217
        STAGE(base + 0);
        // :218 node cross-attention, q = X2, k = v = X3 (after the self-attention, before gcn_3ds): the key | value projection
        // runs on the lane that owns X3, so gcn_3ds below does not have to wait for the 2D side to have read X3
        float* kvx = p->KVx + (size_t)(p->kvx_slots > 1 ? l : 0) * (size_t)N * 2 * D;
        hipEvent_t kvx_ready = nullptr, np2_ready = nullptr;
        if (do2d) {
            RUN(kv_project(h, p, s, h->cross_attn[l], p->X3, kvx));
            if (exact) {
                RUN(after(s, &kvx_ready));                    // (also behind the distance bias)
                RUN(wait(u, kvx_ready));
                RUN(wait(u, x2_ready));
            } else {
                RUN(join());                                  // X2 / E2 of the previous stage are complete
                if (pair && l == 0) RUN(wait(s, x2_ready));   // (the adapter ran on t)
            }
            RUN(attn_block(h, p, exact ? u : s, h->cross_attn[l], p->X2, false, p->Q2n, kvx, p->On2));
            if (exact) {                                      // node side of gcn_2ds[l] on the same lane, then the edge lane may go on
                RUN(gcn_node_project(h, p, u, h->gcn2[l], p->X2, sc2));
                RUN(after(u, &np2_ready));
            }
        }
        STAGE(base + 1);
        if (pair) {                                           // :224-225 as launches of two problems each, behind the previous layer's edge attention
            RUN(wait(s, edge_done));
            RUN(gcn_block_pair(h, p, s, h->gcn3[l], h->gcn2[l], p->X3, p->X2, p->E3, p->E2, e3_pending_relu, inter, sc3, sc2));
            RUN(order(s, t));                                 // t: this layer's edge cross-attention; s goes on with the next layer's node attentions
        } else {
        RUN(fork());                                          // t: gcn_2ds + query projection; s: gcn_3ds + key/value projection
        RUN(gcn_block(h, p, s, h->gcn3[l], p->X3, p->E3, e3_pending_relu, inter, sc3)); // :224
        STAGE(base + 2);
        if (do2d) {
            RUN(wait(t, np2_ready));
            RUN(gcn_block(h, p, t, h->gcn2[l], p->X2, p->E2, 0, inter, sc2, exact));      // :225
            if (exact) RUN(after(t, &x2_ready));
        }
        }
        STAGE(base + 3);
        if (do2d) {   // :231 edge cross-attention: q = 2D edges, k = v = 3D edges (pre-activation)
            const AttnW& w = h->cross_rel[l];
            const float sc2e = (1.f / std::sqrt((float)(D / h->H))) * 1.4426950408889634f;   // 1/sqrt(d_k) * log2(e): the attention works in exp2
            const int dh = D / h->H;
            // format of Q / K|V / O: that of the chain when the bf16 attention kernel is built for this head dim and mode, else fp32
            // PA: operand precision of this block (mode 4: single rounding here, split-bf16 everywhere else -- the 3D outputs, which
            // never see this block, keep the split-bf16 accuracy; profiles/r05_probes/precision_mix_study.txt)
            const int PA = h->prec_attn;
            const bool fa16 = PA && h->flash_bf16 && (dh == 64 || (h->flash_heads_bf16 && S && flash_attn_bf16_supports(dh, PA == 3 ? 3 : 1, h->flash_tr, S)));
            // (mode 4: the chain tensors are split pairs, but Q / K|V / O feed a single-rounded attention only -- they travel as half rows,
            //  which puts the attention on its LDS-direct kernel and the out-projection on the 8-phase one)
            const int SA = fa16 ? ((PA == 1 && S == 1 && h->half_fmt && dh == 64) ? 2 : S) : 0;
            // K | V of layer l go to slot l % 2 on the dependency-exact schedule (the 3D lane may be a layer ahead of the attention
            // that reads them: it waits for the reader of the slot's previous content only)
            const int slot = exact ? (l & 1) : 0;
            float* kve = slot ? p->KVe2 : p->KVe;
            hipStream_t fs = (exact || pair) ? t : s;         // lane of the attention itself
            hipStream_t kvs = pair ? t : s;                   // ... and of its key | value projection (paired schedule: the whole chain on t)
            GemmArgs gq = G(p->E2, D, w.wq, D, p->Qe, D, E, D, w.bq);
            gq.a_split = S; gq.c_split = SA;
            if (SA) gq.c_scale = sc2e;                                 // the split-format attention takes Q pre-scaled
            RUN(gemm(h, t, gq, PA));
            if (!pair) RUN(wait(s, flash_done[slot]));
            GemmArgs gkv = G(p->E3, D, w.wkv, D, kve, 2 * D, E, 2 * D, w.bkv);
            gkv.a_split = S; gkv.c_split = SA;
            RUN(gemm(h, kvs, gkv, PA));
            if (pair) {
            } else if (exact) {
                hipEvent_t kve_ready;
                RUN(after(s, &kve_ready));
                RUN(wait(t, kve_ready));
            } else {
                RUN(join());
            }
            {
                Scope sc(h, fs, PC_FLASH, p->flash_flops);
                FlashSplit sp;
                sp.ablate = h->flash_ablate;
                sp.parts = p->fa_parts; sp.krange = p->d_krange; sp.o_part = p->fa_opart; sp.m_part = p->fa_m;
                sp.l_part = p->fa_l; sp.part_stride = (size_t)E * D; sp.rows = E; sp.heads = h->H;
                if (dh != 32 && dh != 64 && dh != 128)   // any other head dim: VALU attention over the scenes' edge ranges (no bias)
                    RUN(launch_node_attn(p->Qe, D, kve, 2 * D, kve + D, 2 * D, p->Oe, D, nullptr, p->d_edge_ptr32, nullptr,
                                         h->edge_scope == 1 ? 1 : p->S, h->edge_scope == 1 ? E : p->max_e, h->H, D / h->H,
                                         1.f / std::sqrt((float)(D / h->H)), fs, h->node_attn_split));
                else if (fa16) {
                    const bool big = SA == 2 && dh == 64 && p->n_tiles_big && h->flash_tr && h->flash_dma == 1 && h->flash_bq_big;
                    if (big) sp.bq = FLASH_BQ_BIG;
                    sp.qg = h->flash_qg;
                    RUN(launch_flash_attn_bf16(p->Qe, D, kve, kve + (SA == 2 ? D / 2 : D), 2 * D, p->Oe, D, big ? p->d_tiles_big : p->d_tiles, big ? p->n_tiles_big : p->n_tiles,
                                               sc2e, PA == 3 ? 3 : 1, h->flash_tr ? (h->flash_dma >= 3 ? h->flash_dma : h->flash_dma ? 1 : 2) : 0, (SA == 2 && h->half_f16) ? 3 : SA, fs, &sp, h->flash_pv_terms, dh));    // (half rows: V starts at byte 2 D)
                }
                else
                    RUN(launch_flash_attn(p->Qe, D, kve, kve + D, 2 * D, p->Oe, D, p->d_tiles, p->n_tiles, sc2e, fs, &sp, dh));   // (head dims 32 / 128: NUM_HEADS 16 / 4)
            }
            if (exact) RUN(after(t, &flash_done[slot]));
            // out-projection + residual, then LayerNorm (+ inter-layer ReLU).  Split format: the GEMM reads O and the
            // residual as hi/lo pairs and writes plain fp32 into the (now dead) Q buffer; the LayerNorm packs E2 again.
            // Split-pair mode: the residual is added by the LayerNorm kernel instead (as an accumulator init it made the
            // out-projection 60 % slower there -- 397 vs 243 us at the bench size -- against +1 KB per row in the LayerNorm).
            float* pre_ln = S ? p->Qe : p->E2;
            // Half rows: the same (the 8-phase GEMM has no fast accumulator-init path).
            const bool ln_resid = ((S == 1 || (S == 2 && !h->gemm_no_p8)) && h->ln_resid) || (S == 2 && h->half_f16);      // (mode 5: always -- no GEMM reads an fp16 residual)
            GemmArgs o = G(p->Oe, D, w.wo, D, pre_ln, D, E, D, w.bo);
            if (!ln_resid) { o.resid = p->E2; o.ldr = D; o.r_split = S; }
            o.a_split = SA;
            // single-rounded attention (half-row O): pre_ln has one reader, the LayerNorm below -- fp16 half rows instead of fp32 (2^-12 in
            // front of a normalisation whose output is rounded to bf16 / split pairs: not visible in the mode's error; half the bytes both ways)
            const bool o16 = ln_resid && SA == 2 && h->outproj_f16 != 0 && D % 256 == 0;
            if (o16) o.c_f16_cols = D;
            RUN(gemm(h, fs, o, PA));
            {
                Scope sc(h, fs, PC_LAYERNORM, 0);
                const int SL = (S == 2 && h->half_f16) ? 4 : S;          // (mode 5: the half rows the LayerNorm reads and writes hold fp16)
                RUN(launch_layernorm_to(pre_ln, D, p->E2, D, E, D, w.lng, w.lnb, inter, SL, fs, ln_resid ? p->E2 : nullptr, D, SL, o16 ? 1 : 0));
            }
            if (pair) RUN(after(t, &edge_done));
        }
        e3_pending_relu = inter;
        STAGE(base + 4);
    }
    if (tr && tr->edge_dis && E > 0) {
        // gcn_edge_feature_2d_dis = triplet_projector_2d(cat[x2[ei[0]], x2[ei[1]], e2]) (:259-264,319-322): node columns of
        // its first Linear on N rows, gathered into the edge GEMM's accumulators like nn_edge.0
        if (!h->trip.wnode) return fail(VLSAT_ESTATE, "forward(istrain=True): triplet_projector_2d weights were not loaded");
        const int NPC = npc_of(h);
        RUN(gemm(h, s, G(p->X2, LDX, h->trip.wnode, D, sc2.NP, NPC, N, 4 * D, h->trip.bnode)));
        GemmArgs e1 = G(p->E2, D, h->trip.we, D, sc2.Hbig, 2 * D, E, 2 * D, nullptr, ACT_RELU);
        e1.g0 = sc2.NP; e1.gi0 = p->d_src; e1.ldg0 = NPC;
        e1.g1 = sc2.NP + 2 * D; e1.gi1 = p->d_dst; e1.ldg1 = NPC;
        e1.a_split = S; e1.c_split = S;
        RUN(gemm(h, s, e1));
        GemmArgs e2 = G(sc2.Hbig, 2 * D, h->trip.w2, 2 * D, tr->edge_dis, D, E, D, h->trip.b2);
        e2.a_split = S;
        RUN(gemm(h, s, e2));
    }
    // a-15 relation heads, a-16 object heads
    if (pair) {                                               // both head pairs as launches of two problems each, behind the last edge attention
        RUN(wait(s, edge_done));
        RUN(rel_head_pair(h, p, s, p->E3, e3_pending_relu, rel3d, p->E2, rel2d, sc3, sc2));
        RUN(obj_head_pair(h, p, s, obj3d, obj2d, sc3, sc2));
    } else {
    RUN(fork());                                              // t: the 2D heads
    if (E > 0) {
        RUN(rel_head(h, p, s, h->rel3, p->E3, e3_pending_relu, rel3d, sc3));
        if (do2d) RUN(rel_head(h, p, t, h->rel2, p->E2, 0, rel2d, sc2));
    }
    RUN(obj_head(h, p, s, p->X3, h->obj3_w, h->obj3_b, obj3d, sc3));
    if (do2d) {
        RUN(wait(u, x2_ready));
        RUN(obj_head(h, p, u, p->X2, h->obj2_w, h->obj2_b, obj2d, sc2));
    }
    }
    if (dual) {                                               // every lane ends joined into the caller's stream
        profile_close(h, t);
        profile_close(h, u);
        RUN(order(t, s));
        if (u != t) RUN(order(u, s));
    }
    profile_close(h, s);
    VLSAT_HIP_CHECK(hipEventRecord(p->last_use, s));
#undef STAGE
    return 0;
}

int vlsat_forward(vlsat_handle h, vlsat_plan p, const float* pts, const float* f2d, const float* desc,
                  float* obj3d, float* obj2d, float* rel3d, float* rel2d, void* stream) {
    return forward_impl(h, p, pts, f2d, desc, obj3d, obj2d, rel3d, rel2d, nullptr, stream);
}

// Mmgnet.forward(istrain=True) without autograd: the four eval outputs plus obj_feature_3d_mimic [N,512],
// obj_features_2d_mimic [N,512] and gcn_edge_feature_2d_dis [E,512] (reference SGFN_MMG/model.py:332-333); the eighth
// element of the reference's tuple, exp(obj_logit_scale), is a constant of VlsatDims.
int vlsat_forward_train(vlsat_handle h, vlsat_plan p, const float* pts, const float* f2d, const float* desc,
                        float* obj3d, float* obj2d, float* rel3d, float* rel2d, float* obj_feature_3d_mimic,
                        float* obj_features_2d_mimic, float* gcn_edge_feature_2d_dis, void* stream) {
    if (!obj2d || !f2d) return fail(VLSAT_EINVAL, "vlsat_forward_train: the 2D branch is required");
    if (!obj_feature_3d_mimic || !obj_features_2d_mimic || (p && p->E > 0 && !gcn_edge_feature_2d_dis))
        return fail(VLSAT_EINVAL, "vlsat_forward_train: null output");
    const TrainOut tr{obj_feature_3d_mimic, obj_features_2d_mimic, gcn_edge_feature_2d_dis};
    return forward_impl(h, p, pts, f2d, desc, obj3d, obj2d, rel3d, rel2d, &tr, stream);
}

int vlsat_profile_enable(vlsat_handle h, int32_t enable) {
    if (!h) return fail(VLSAT_EINVAL, "null handle");
    h->prof = enable != 0;
    return 0;
}
int vlsat_profile_num_classes(void) { return PC_COUNT; }
const char* vlsat_profile_class_name(int32_t c) { return (c >= 0 && c < PC_COUNT) ? kProfNames[c] : ""; }

static int profile_drain(vlsat_handle h) {
    if (h->recs.empty()) return 0;
    struct Iv { float a, b; };
    std::vector<Iv> iv[PC_COUNT];
    for (auto& r : h->recs) {
        VLSAT_HIP_CHECK(hipEventSynchronize(r.b));
        Iv x{0.f, 0.f};
        VLSAT_HIP_CHECK(hipEventElapsedTime(&x.a, h->prof_base, r.a));
        VLSAT_HIP_CHECK(hipEventElapsedTime(&x.b, h->prof_base, r.b));
        iv[r.cls].push_back(x);
        h->acc_n[r.cls] += r.kernels;
        h->acc_fl[r.cls] += r.flops;
    }
    for (int c = 0; c < PC_COUNT; ++c) {                 // length of the union of the class's intervals (two streams may overlap)
        auto& v = iv[c];
        std::sort(v.begin(), v.end(), [](const Iv& x, const Iv& y) { return x.a < y.a; });
        double tot = 0, end = -1e30;
        for (const Iv& x : v) {
            if (x.a > end) { tot += x.b - x.a; end = x.b; }
            else if (x.b > end) { tot += x.b - end; end = x.b; }
        }
        h->acc_ms[c] += tot;
    }
    h->recs.clear();
    h->ev_used = 0;
    h->prof_base_set = false;
    return 0;
}
int vlsat_profile_read(vlsat_handle h, int32_t cls, double* total_ms, int64_t* launches, double* flops) {
    if (!h || cls < 0 || cls >= PC_COUNT) return fail(VLSAT_EINVAL, "bad profile class");
    RUN(profile_drain(h));
    if (total_ms) *total_ms = h->acc_ms[cls];
    if (launches) *launches = h->acc_n[cls];
    if (flops) *flops = h->acc_fl[cls];
    h->acc_ms[cls] = 0; h->acc_n[cls] = 0; h->acc_fl[cls] = 0;
    return 0;
}

}  // extern "C"
