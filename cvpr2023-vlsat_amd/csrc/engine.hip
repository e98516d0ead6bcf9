// Host engine of libvlsat_hip.so: weight preparation, graph plan, forward orchestration and
// the C ABI declared in include/vlsat.h.  All device work is the hand-written kernels of
// this directory; there is no CPU fallback anywhere in this library.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/vlsat.h"
#include "common.h"
#include "kernels.h"

namespace vlsat {

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(int code, const std::string& m) {
    g_err = m;
    return code;
}

// ------------------------------------------------------------------------------------------
enum ProfClass { PC_GEMM = 0, PC_FLASH, PC_POINTNET, PC_GATE, PC_NODE_ATTN, PC_LAYERNORM, PC_AGGREGATE, PC_MISC, PC_COUNT };
static const char* kProfNames[PC_COUNT] = {"gemm_f32", "flash_attn_f32", "pointnet", "edge_gate", "node_attn",
                                           "layernorm512", "aggregate", "misc"};

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
};

struct AttnW {          // one MultiHeadAttention block
    float *wq, *bq, *wkv, *bkv, *wo, *bo, *lng, *lnb;
    float *wqkv, *bqkv;  // self-attention: fused [1536,512]
};
struct GcnW {           // one GraphEdgeAttenNetwork block
    float *wnode, *bnode;   // [3328,512]: Wi | Wj | Wgq | Wv
    float *we1;             // [1024,512] edge part of nn_edge.0
    float *we2, *be2;       // nn_edge.2
    float *wpe, *bpe;       // proj_edge, rows permuted head-major
    float *w0k, *w3, *b3;   // gate MLP
    float *wp0, *bp0, *wp2, *bp2;
};
struct RelHeadW { float *w1, *b1, *w2, *b2, *w3, *b3; };
// STNkd(k=64) with its five BatchNorm1d(eval) layers folded and the identity folded into the last bias
struct StnW { float *c1, *c1b, *c2, *c2b, *c3, *c3b, *f1, *f1b, *f2, *f2b, *f3, *f3b; };

}  // namespace vlsat

using namespace vlsat;

struct vlsat_ctx {
    VlsatDims d{};
    int dual_stream = 1;     // run the 2D twin stages of small plans on a second stream (VLSAT_DUAL_STREAM=0 disables)
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> sync_ev;     // fork/join events (timing disabled), created on first use
    int fa_split = 1;        // allow the split-key edge attention for small plans (VLSAT_FLASH_SPLIT=0 disables)
    int edge_scope = 0;      // edge cross-attention keys: 0 = the query's scene, 1 = the whole batch (vlsat_set_edge_attention_scope)
    int D = 512, A = 256, H = 8, C_pt = 768;
    std::map<std::string, std::vector<float>> host;   // raw reference-layout tensors
    bool finalized = false;
    std::vector<float*> dev_allocs;
    // prepared device weights
    float *pn_w1, *pn_b1, *pn_w2, *pn_b2, *pn_w3, *pn_b3;
    float *mlp_w, *mlp_b;
    float *re_w1cat, *re_b1cat;
    float *re3_w2, *re3_b2, *re3_w3, *re3_b3, *re2_w2, *re2_b2, *re2_w3, *re2_b3;
    float *ad_w1, *ad_b1, *ad_w2h, *ad_b2h;
    DistBiasW db{};
    std::vector<AttnW> self_attn, cross_attn, cross_rel;
    std::vector<GcnW> gcn3, gcn2;
    RelHeadW rel3{}, rel2{};
    StnW stn_obj{}, stn_re3{}, stn_re2{};          // MODEL.feature_transform
    float *obj3_w, *obj3_b, *obj2_w, *obj2_b;
    // profiling
    bool prof = false;
    struct Rec { int cls; hipEvent_t a, b; double flops; long kernels; };
    std::vector<Rec> recs;
    Rec open{};              // interval of the kernel class currently being launched (see Scope)
    bool open_ok = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double acc_ms[PC_COUNT] = {0};
    int64_t acc_n[PC_COUNT] = {0};
    double acc_fl[PC_COUNT] = {0};
    int debug_stop = -1;
    // GEMM operand precision: 0 exact fp32 MFMA, 1 bf16, 3 split-bf16 (vlsat_set_gemm_precision)
    int prec = 0;
    std::map<const float*, std::pair<uint16_t*, uint16_t*>> split;
    // workspace arenas of destroyed plans, re-used by the next plan that fits (an eval loop
    // builds one plan per scene; hipMalloc/hipFree per scene would dominate small scenes)
    std::vector<std::pair<char*, size_t>> arena_pool;
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
    // device index arrays
    int32_t *d_src, *d_dst, *d_rowptr, *d_order, *d_scene_ptr;
    int64_t* d_bias_ptr;
    int4* d_tiles;
    int n_tiles = 0;
    // split-key mode of the edge attention for plans with too few blocks to fill the chip (flash_attn_f32.hip)
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
    float* stn_ws = nullptr;                        // MODEL.feature_transform scratch (carved per phase in stn_phase)
    size_t stn_ws_floats = 0;
    bool dual = false;
    float *NP2 = nullptr, *Hbig2 = nullptr, *KP2 = nullptr, *G2 = nullptr, *T768b = nullptr, *rs2 = nullptr, *H2b = nullptr;
};

// scratch buffers of one modality branch
struct Scratch { float *NP, *Hbig, *KP, *G, *T768, *R1, *R2, *rs, *H2; };
static Scratch scratch_of(const vlsat_plan_s* p, int branch) {
    if (branch == 1 && p->dual)
        return {p->NP2, p->Hbig2, p->KP2, p->G2, p->T768b, p->Hbig2, p->Hbig2 + (size_t)std::max<int64_t>(p->E, 1) * 512, p->rs2, p->H2b};
    return {p->NP, p->Hbig, p->KP, p->G, p->T768, p->R1, p->R2, p->rs, p->H2};
}

namespace {

const std::vector<float>* find(vlsat_ctx* h, const std::string& k) {
    auto it = h->host.find(k);
    return it == h->host.end() ? nullptr : &it->second;
}

int upload(vlsat_ctx* h, const std::vector<float>& v, float** out) {
    float* p = nullptr;
    VLSAT_HIP_CHECK(hipMalloc(&p, std::max<size_t>(v.size(), 4) * sizeof(float)));
    VLSAT_HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    h->dev_allocs.push_back(p);
    *out = p;
    return 0;
}

struct Prep {
    vlsat_ctx* h;
    std::string missing;
    const std::vector<float>& get(const std::string& k, size_t expect) {
        static const std::vector<float> empty;
        auto* v = find(h, k);
        if (!v || v->size() != expect) {
            if (missing.empty()) missing = k + (v ? " (wrong element count)" : "");
            return empty;
        }
        return *v;
    }
};

#define UP(vec, dst)                                 \
    do {                                             \
        int _r = upload(h, (vec), &(dst));           \
        if (_r) return _r;                           \
    } while (0)

int prepare_attn(vlsat_ctx* h, Prep& P, const std::string& pre, AttnW& w, bool fuse_qkv, float qscale) {
    const size_t D = h->D;
    auto wq = P.get(pre + ".attention.fc_q.weight", D * D), bq = P.get(pre + ".attention.fc_q.bias", D);
    auto wk = P.get(pre + ".attention.fc_k.weight", D * D), bk = P.get(pre + ".attention.fc_k.bias", D);
    auto wv = P.get(pre + ".attention.fc_v.weight", D * D), bv = P.get(pre + ".attention.fc_v.bias", D);
    auto wo = P.get(pre + ".attention.fc_o.weight", D * D), bo = P.get(pre + ".attention.fc_o.bias", D);
    auto g = P.get(pre + ".layer_norm.weight", D), b = P.get(pre + ".layer_norm.bias", D);
    if (!P.missing.empty()) return 0;
    for (auto& x : wq) x *= qscale;   // 1/sqrt(d_k) = 0.125 is a power of two: exact
    for (auto& x : bq) x *= qscale;
    std::vector<float> wkv(wk), bkv(bk);
    wkv.insert(wkv.end(), wv.begin(), wv.end());
    bkv.insert(bkv.end(), bv.begin(), bv.end());
    UP(wq, w.wq); UP(bq, w.bq); UP(wkv, w.wkv); UP(bkv, w.bkv); UP(wo, w.wo); UP(bo, w.bo); UP(g, w.lng); UP(b, w.lnb);
    w.wqkv = w.bqkv = nullptr;
    if (fuse_qkv) {
        std::vector<float> wqkv(wq), bqkv(bq);
        wqkv.insert(wqkv.end(), wkv.begin(), wkv.end());
        bqkv.insert(bqkv.end(), bkv.begin(), bkv.end());
        UP(wqkv, w.wqkv); UP(bqkv, w.bqkv);
    }
    return 0;
}

int prepare_gcn(vlsat_ctx* h, Prep& P, const std::string& pre, GcnW& w) {
    const int D = h->D, A = h->A, H = h->H;
    const int dn = D / H, de = D / H, dox = A / H;   // 64, 64, 32
    const std::string e = pre + ".edgeatten.";
    auto w_e0 = P.get(e + "nn_edge.0.weight", (size_t)2 * D * 3 * D), b_e0 = P.get(e + "nn_edge.0.bias", 2 * D);
    auto w_e2 = P.get(e + "nn_edge.2.weight", (size_t)D * 2 * D), b_e2 = P.get(e + "nn_edge.2.bias", D);
    // gate MLP input: cat[q, k] (USE_GCN_EDGE, dn+de columns) or q alone (dn columns, hidden width 2*dn); reference
    // network_MMG.py:72-75
    const int NIN = h->d.use_gcn_edge ? dn + de : dn;
    if (!h->d.use_gcn_edge && 2 * dn != dn + de) return fail(VLSAT_EINVAL, "USE_GCN_EDGE=false needs d_n == d_e");
    auto w_n0 = P.get(e + "nn.0.weight", (size_t)(dn + de) * NIN), b_n0 = P.get(e + "nn.0.bias", dn + de);
    auto w_n3 = P.get(e + "nn.3.weight", (size_t)dox * (dn + de)), b_n3 = P.get(e + "nn.3.bias", dox);
    auto w_pe = P.get(e + "proj_edge.0.weight", (size_t)D * D), b_pe = P.get(e + "proj_edge.0.bias", D);
    auto w_pq = P.get(e + "proj_query.0.weight", (size_t)D * D), b_pq = P.get(e + "proj_query.0.bias", D);
    auto w_pv = P.get(e + "proj_value.0.weight", (size_t)A * D), b_pv = P.get(e + "proj_value.0.bias", A);
    auto w_p0 = P.get(pre + ".prop.0.weight", (size_t)(D + A) * (D + A)), b_p0 = P.get(pre + ".prop.0.bias", D + A);
    auto w_p2 = P.get(pre + ".prop.2.weight", (size_t)D * (D + A)), b_p2 = P.get(pre + ".prop.2.bias", D);
    if (!P.missing.empty()) return 0;
    if (dn != 64 || de != 64 || dox != 32) return fail(VLSAT_EINVAL, "gate kernel is built for 8 heads x (64,64,32)");

    // nn_edge.0 [1024, 1536] column blocks: [0:512] = x_i (source), [512:1024] = edge, [1024:1536] = x_j (target)
    const int NO = 2 * D, NI = 3 * D;
    const int NODE_COLS = 2 * NO + H * (dn + de) + A;       // 1024 + 1024 + 1024 + 256 = 3328
    std::vector<float> wnode((size_t)NODE_COLS * D, 0.f), bnode(NODE_COLS, 0.f), we1((size_t)NO * D);
    for (int o = 0; o < NO; ++o) {
        const float* r = &w_e0[(size_t)o * NI];
        std::memcpy(&wnode[(size_t)o * D], r, D * sizeof(float));
        std::memcpy(&we1[(size_t)o * D], r + D, D * sizeof(float));
        std::memcpy(&wnode[(size_t)(NO + o) * D], r + 2 * D, D * sizeof(float));
        bnode[o] = b_e0[o];
    }
    // Gq[h*128 + o] = sum_c W0[o, c] * q[c*8 + h] + b0[o],  q = proj_query(x): fold into one [1024,512] matrix
    const int G0 = 2 * NO, HID = dn + de;   // 128
    for (int hh = 0; hh < H; ++hh)
        for (int o = 0; o < HID; ++o) {
            std::vector<double> row(D, 0.0);
            double bb = b_n0[o];
            for (int c = 0; c < dn; ++c) {
                const double wc = w_n0[(size_t)o * NIN + c];
                const float* qrow = &w_pq[(size_t)(c * H + hh) * D];
                for (int k = 0; k < D; ++k) row[k] += wc * qrow[k];
                bb += wc * b_pq[c * H + hh];
            }
            float* dst = &wnode[(size_t)(G0 + hh * HID + o) * D];
            for (int k = 0; k < D; ++k) dst[k] = (float)row[k];
            bnode[G0 + hh * HID + o] = (float)bb;
        }
    // value rows HEAD-MAJOR: row h*dox + m <- proj_value row m*H + h, so that the gate kernel reads / writes the four
    // consecutive channels a lane owns as one float4.  The gated and aggregated tensors inherit that channel order
    // (max / add / mean are per channel) and the columns of prop.0 that read them are permuted to match below.
    const int V0 = G0 + H * HID;
    for (int hh = 0; hh < H; ++hh)
        for (int m = 0; m < dox; ++m) {
            std::memcpy(&wnode[(size_t)(V0 + hh * dox + m) * D], &w_pv[(size_t)(m * H + hh) * D], D * sizeof(float));
            bnode[V0 + hh * dox + m] = b_pv[m * H + hh];
        }
    {
        std::vector<float> perm(w_p0.size());
        const int IN = D + A;
        for (int o = 0; o < IN; ++o) {
            std::memcpy(&perm[(size_t)o * IN], &w_p0[(size_t)o * IN], D * sizeof(float));
            for (int hh = 0; hh < H; ++hh)
                for (int m = 0; m < dox; ++m) perm[(size_t)o * IN + D + hh * dox + m] = w_p0[(size_t)o * IN + D + m * H + hh];
        }
        w_p0.swap(perm);
    }
    // proj_edge rows permuted: row h*64 + c <- original row c*8 + h
    std::vector<float> wpe((size_t)D * D), bpe(D);
    for (int hh = 0; hh < H; ++hh)
        for (int c = 0; c < de; ++c) {
            std::memcpy(&wpe[(size_t)(hh * de + c) * D], &w_pe[(size_t)(c * H + hh) * D], D * sizeof(float));
            bpe[hh * de + c] = b_pe[c * H + hh];
        }
    std::vector<float> w0k((size_t)HID * de, 0.f);           // edge half of layer 1 (unused without USE_GCN_EDGE)
    if (h->d.use_gcn_edge)
        for (int o = 0; o < HID; ++o)
            for (int c = 0; c < de; ++c) w0k[(size_t)o * de + c] = w_n0[(size_t)o * NIN + dn + c];
    UP(wnode, w.wnode); UP(bnode, w.bnode); UP(we1, w.we1); UP(w_e2, w.we2); UP(b_e2, w.be2);
    UP(wpe, w.wpe); UP(bpe, w.bpe); UP(w0k, w.w0k); UP(w_n3, w.w3); UP(b_n3, w.b3);
    UP(w_p0, w.wp0); UP(b_p0, w.bp0); UP(w_p2, w.wp2); UP(b_p2, w.bp2);
    return 0;
}

// STNkd weights of encoder `enc` (reference network_PointNet.py:52-86): conv/fc + BatchNorm1d(eval) folded in fp64,
// "+ eye(64)" folded into fc3's bias
int prepare_stn(vlsat_ctx* h, Prep& P, const std::string& enc, StnW& w) {
    const std::string f = enc + ".fstn.";
    struct L { const char* name; const char* bn; int out, in; };
    const L layers[6] = {{"conv1", "bn1", 64, 64}, {"conv2", "bn2", 128, 64}, {"conv3", "bn3", 1024, 128},
                         {"fc1", "bn4", 512, 1024}, {"fc2", "bn5", 256, 512}, {"fc3", nullptr, 4096, 256}};
    float** dst[6][2] = {{&w.c1, &w.c1b}, {&w.c2, &w.c2b}, {&w.c3, &w.c3b}, {&w.f1, &w.f1b}, {&w.f2, &w.f2b}, {&w.f3, &w.f3b}};
    for (int i = 0; i < 6; ++i) {
        const L& l = layers[i];
        auto wt = P.get(f + l.name + ".weight", (size_t)l.out * l.in), bs = P.get(f + l.name + ".bias", l.out);
        if (l.bn) {
            const std::string b = f + l.bn;
            auto g = P.get(b + ".weight", l.out), be = P.get(b + ".bias", l.out);
            auto mu = P.get(b + ".running_mean", l.out), var = P.get(b + ".running_var", l.out);
            if (!P.missing.empty()) return 0;
            for (int o = 0; o < l.out; ++o) {
                const double sc = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
                for (int k = 0; k < l.in; ++k) wt[(size_t)o * l.in + k] = (float)(wt[(size_t)o * l.in + k] * sc);
                bs[o] = (float)(((double)bs[o] - mu[o]) * sc + be[o]);
            }
        } else {
            if (!P.missing.empty()) return 0;
            for (int d = 0; d < 64; ++d) bs[d * 64 + d] += 1.f;
        }
        UP(wt, *dst[i][0]);
        UP(bs, *dst[i][1]);
    }
    return 0;
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
// Per-class timing with as few events as possible: an event is recorded only where the kernel CLASS changes
// (a run of consecutive launches of one class is one interval), plus one at the end of the forward.
struct Scope {
    vlsat_ctx* h;
    hipStream_t s;
    int cls;
    double flops;
    long k0 = 0;
    Scope(vlsat_ctx* h_, hipStream_t s_, int cls_, double fl) : h(h_), s(s_), cls(cls_), flops(fl) {
        k0 = gemm_kernel_launches();
        if (!h->prof) return;
        if (h->open_ok && h->open.cls == cls) return;              // same class: the open interval continues
        hipEvent_t e = next_event(h);
        hipEventRecord(e, s);
        if (h->open_ok) { h->open.b = e; h->recs.push_back(h->open); }
        h->open = {cls, e, e, 0.0, 0};
        h->open_ok = true;
    }
    ~Scope() {
        if (!h->prof) return;
        h->open.flops += flops;
        h->open.kernels += cls == PC_GEMM ? gemm_kernel_launches() - k0 : 1;
    }
};
// end of a forward (or of a debug-stopped one): close the open interval
static void profile_close(vlsat_ctx* h, hipStream_t s) {
    if (!h->prof || !h->open_ok) return;
    hipEvent_t e = next_event(h);
    hipEventRecord(e, s);
    h->open.b = e;
    h->recs.push_back(h->open);
    h->open_ok = false;
}

int gemm(vlsat_ctx* h, hipStream_t s, const GemmArgs& a0) {
    GemmArgs a = a0;
    if (h->prec) {
        // split-bf16 GEMM path: weights are split into bf16 hi/lo parts once, on first use
        // (i.e. during warm-up), keyed by the fp32 weight pointer
        auto it = h->split.find(a.W);
        if (it == h->split.end()) {
            const size_t n = (size_t)a.N * a.K;
            uint16_t *hi = nullptr, *lo = nullptr;
            VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&hi), n * 2 + 16));
            VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&lo), n * 2 + 16));
            if (int r = launch_split_bf16(a.W, n, hi, lo, s)) return r;
            it = h->split.emplace(a.W, std::make_pair(hi, lo)).first;
        }
        a.prec = h->prec;
        a.Whi = it->second.first;
        a.Wlo = it->second.second;
    }
    Scope sc(h, s, PC_GEMM, gemm_flops(a));
    return launch_gemm(a, s);
}

GemmArgs G(const float* A, int lda, const float* W, int K, float* C, int ldc, int M, int N, const float* bias,
           int act = ACT_NONE) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = act;
    return g;
}

#define RUN(expr)                  \
    do {                           \
        int _r = (expr);           \
        if (_r) return _r;         \
    } while (0)

int attn_block(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const AttnW& w, float* xq, const float* xkv, bool self) {
    const int N = (int)p->N, D = h->D, LDX = 768;
    if (self) {
        RUN(gemm(h, s, G(xq, LDX, w.wqkv, D, p->QKVn, 3 * D, N, 3 * D, w.bqkv)));
    } else {
        RUN(gemm(h, s, G(xq, LDX, w.wq, D, p->QKVn, 3 * D, N, D, w.bq)));
        RUN(gemm(h, s, G(xkv, LDX, w.wkv, D, p->QKVn + D, 3 * D, N, 2 * D, w.bkv)));
    }
    {
        Scope sc(h, s, PC_NODE_ATTN, 0);
        RUN(launch_node_attn(p->QKVn, 3 * D, p->QKVn + D, 3 * D, p->QKVn + 2 * D, 3 * D, p->On, D, p->bias,
                             p->d_scene_ptr, p->d_bias_ptr, p->S, p->max_n, h->H, 1.0f, s));
    }
    GemmArgs o = G(p->On, D, w.wo, D, xq, LDX, N, D, w.bo);
    o.resid = xq; o.ldr = LDX;
    RUN(gemm(h, s, o));
    {
        Scope sc(h, s, PC_LAYERNORM, 0);
        RUN(launch_layernorm(xq, LDX, N, D, w.lng, w.lnb, 0, s));
    }
    return 0;
}

int gcn_block(vlsat_ctx* h, vlsat_plan_s* p, hipStream_t s, const GcnW& w, float* x, float* e, int e_relu_pending,
              int out_relu, const Scratch& sc) {
    const int N = (int)p->N, E = (int)p->E, D = h->D, A = h->A, LDX = 768, NPC = 3328;
    RUN(gemm(h, s, G(x, LDX, w.wnode, D, sc.NP, NPC, N, NPC, w.bnode)));
    GemmArgs e1 = G(e, D, w.we1, D, sc.Hbig, 2 * D, E, 2 * D, nullptr, ACT_RELU);
    e1.relu_a = e_relu_pending;
    e1.g0 = sc.NP; e1.gi0 = p->d_src; e1.ldg0 = NPC;
    e1.g1 = sc.NP + 2 * D; e1.gi1 = p->d_dst; e1.ldg1 = NPC;
    RUN(gemm(h, s, e1));
    if (h->d.use_gcn_edge) {              // proj_edge feeds only the gate MLP (reference network_MMG.py:98-102)
        GemmArgs kp = G(e, D, w.wpe, D, sc.KP, D, E, D, w.bpe);
        kp.relu_a = e_relu_pending;
        RUN(gemm(h, s, kp));
    }
    RUN(gemm(h, s, G(sc.Hbig, 2 * D, w.we2, 2 * D, e, D, E, D, w.be2)));   // e <- nn_edge output (pre-activation)
    {
        GateArgs g{};
        g.kproj = sc.KP; g.node = sc.NP; g.ld_node = NPC; g.gq_off = 4 * D; g.v_off = 4 * D + h->H * 128;
        g.src = p->d_src; g.dst = p->d_dst; g.w0k = w.w0k; g.w3 = w.w3; g.b3 = w.b3; g.gated = sc.G;
        g.prob = p->prob; g.n_edges = E; g.use_edge = h->d.use_gcn_edge;
        Scope scope(h, s, PC_GATE, (double)E * h->H * (2.0 * 64 * 128 + 2.0 * 128 * 32));
        RUN(launch_edge_gate(g, s));
    }
    {
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
    GemmArgs a = G(e, D, w.w1, D, sc.R1, 512, E, 512, w.b1, ACT_RELU);
    a.relu_a = relu_a;
    RUN(gemm(h, s, a));
    RUN(gemm(h, s, G(sc.R1, 512, w.w2, 512, sc.R2, 256, E, 256, w.b2, ACT_RELU)));
    // multi_rel_outputs: sigmoid (PointNetRelClsMulti) or log_softmax over the R classes (PointNetRelCls)
    RUN(gemm(h, s, G(sc.R2, 256, w.w3, 256, out, R, E, R, w.b3, h->d.multi_rel_outputs ? ACT_SIGMOID : ACT_NONE)));
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
        RUN(launch_row_invnorm(x, 768, N, D, std::exp(h->d.obj_logit_scale), sc.rs, s));
    }
    GemmArgs a = G(x, 768, w, D, out, C, N, C, b);
    a.rowscale = sc.rs;
    RUN(gemm(h, s, a));
    return 0;
}

}  // namespace

// ============================================================================================
extern "C" {

const char* vlsat_last_error(void) { return g_err.c_str(); }
const char* vlsat_version(void) { return "vlsat-hip gfx950 fp32-mfma r1"; }

int vlsat_create(const VlsatDims* d, vlsat_handle* out) {
    if (!d || !out) return fail(VLSAT_EINVAL, "vlsat_create: null argument");
    if (d->n_layers < 1 || d->n_layers > 16) return fail(VLSAT_EINVAL, "n_layers must be in [1,16]");
    if (d->n_heads != 8 || d->dim_atten != 256) return fail(VLSAT_EINVAL, "only NUM_HEADS=8, DIM_ATTEN=256 are built");
    if (d->gcn_aggr < 0 || d->gcn_aggr > 2) return fail(VLSAT_EINVAL, "gcn_aggr must be 0 (max), 1 (add) or 2 (mean)");
    if (d->dim_point != 3 && d->dim_point != 6 && d->dim_point != 9)
        return fail(VLSAT_EINVAL, "dim_point must be 3, 6 or 9 (xyz [+ USE_RGB] [+ USE_NORMAL])");
    if (d->feature_transform != 0 && d->feature_transform != 1) return fail(VLSAT_EINVAL, "feature_transform must be 0 or 1");
    if (d->n_obj_class < 1 || d->n_rel_class < 1) return fail(VLSAT_EINVAL, "class counts must be positive");
    auto* h = new (std::nothrow) vlsat_ctx();
    if (!h) return fail(VLSAT_ENOMEM, "out of host memory");
    h->d = *d;
    if (const char* e = getenv("VLSAT_FLASH_SPLIT")) h->fa_split = atoi(e);
    if (const char* e = getenv("VLSAT_DUAL_STREAM")) h->dual_stream = atoi(e);
    *out = h;
    return 0;
}

void vlsat_destroy(vlsat_handle h) {
    if (!h) return;
    for (float* p : h->dev_allocs) hipFree(p);
    for (auto& a : h->arena_pool) hipFree(a.first);
    for (auto& kv : h->split) { hipFree(kv.second.first); hipFree(kv.second.second); }
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    for (hipEvent_t e : h->sync_ev) hipEventDestroy(e);
    if (h->side) hipStreamDestroy(h->side);
    delete h;
}

int vlsat_load_weight(vlsat_handle h, const char* name, const float* host, size_t count) {
    if (!h || !name || !host) return fail(VLSAT_EINVAL, "vlsat_load_weight: null argument");
    if (h->finalized) return fail(VLSAT_ESTATE, "weights already finalised");
    std::string k(name);
    static const char* prefixes[] = {"obj_encoder.", "rel_encoder_2d.", "rel_encoder_3d.", "mlp_3d.", "clip_adapter.fc",
                                     "mmg.", "rel_predictor_3d.", "rel_predictor_2d.", "obj_predictor_3d.",
                                     "obj_predictor_2d."};
    bool ok = false;
    for (auto p : prefixes) ok |= k.rfind(p, 0) == 0;
    if (!ok) return fail(VLSAT_EINVAL, "unknown weight name: " + k);
    h->host[k].assign(host, host + count);
    return 0;
}

int vlsat_finalize_weights(vlsat_handle h) {
    if (!h) return fail(VLSAT_EINVAL, "null handle");
    if (h->finalized) return 0;
    Prep P{h, ""};
    const int D = h->D, C = h->C_pt, L = h->d.n_layers;
    // object encoder
    auto w1 = P.get("obj_encoder.conv1.weight", 64 * (size_t)h->d.dim_point), b1 = P.get("obj_encoder.conv1.bias", 64);
    auto w2 = P.get("obj_encoder.conv2.weight", 128 * 64), b2 = P.get("obj_encoder.conv2.bias", 128);
    auto w3 = P.get("obj_encoder.conv3.weight", (size_t)C * 128), b3 = P.get("obj_encoder.conv3.bias", C);
    // mlp_3d with BatchNorm1d(eval) folded (reference SGFN_MMG/model.py:106-111)
    const int M3 = D - 8;
    auto mw = P.get("mlp_3d.0.weight", (size_t)M3 * C), mb = P.get("mlp_3d.0.bias", M3);
    auto bg = P.get("mlp_3d.1.weight", M3), bb = P.get("mlp_3d.1.bias", M3);
    auto bm = P.get("mlp_3d.1.running_mean", M3), bv = P.get("mlp_3d.1.running_var", M3);
    if (P.missing.empty()) {
        for (int o = 0; o < M3; ++o) {
            const double sc = (double)bg[o] / std::sqrt((double)bv[o] + 1e-5);
            for (int k = 0; k < C; ++k) mw[(size_t)o * C + k] = (float)(mw[(size_t)o * C + k] * sc);
            mb[o] = (float)(((double)mb[o] - bm[o]) * sc + bb[o]);
        }
    }
    // relation encoders
    std::vector<float> w1cat, b1cat;
    std::vector<float> r3w2, r3b2, r3w3, r3b3, r2w2, r2b2, r2w3, r2b3;
    for (const char* br : {"rel_encoder_3d", "rel_encoder_2d"}) {
        std::string b(br);
        auto c1 = P.get(b + ".conv1.weight", 64 * 11), c1b = P.get(b + ".conv1.bias", 64);
        w1cat.insert(w1cat.end(), c1.begin(), c1.end());
        b1cat.insert(b1cat.end(), c1b.begin(), c1b.end());
        auto c2 = P.get(b + ".conv2.weight", 128 * 64), c2b = P.get(b + ".conv2.bias", 128);
        auto c3 = P.get(b + ".conv3.weight", (size_t)D * 128), c3b = P.get(b + ".conv3.bias", D);
        if (b == "rel_encoder_3d") { r3w2 = c2; r3b2 = c2b; r3w3 = c3; r3b3 = c3b; }
        else { r2w2 = c2; r2b2 = c2b; r2w3 = c3; r2b3 = c3b; }
    }
    if (h->d.feature_transform) {
        RUN(prepare_stn(h, P, "obj_encoder", h->stn_obj));
        RUN(prepare_stn(h, P, "rel_encoder_3d", h->stn_re3));
        RUN(prepare_stn(h, P, "rel_encoder_2d", h->stn_re2));
    }
    // adapter: 0.5*(W2 h + b2) + 0.5*x  -> halve W2,b2 (exact), residual scale 0.5
    auto aw1 = P.get("clip_adapter.fc1.weight", 256 * (size_t)D), ab1 = P.get("clip_adapter.fc1.bias", 256);
    auto aw2 = P.get("clip_adapter.fc2.weight", (size_t)D * 256), ab2 = P.get("clip_adapter.fc2.bias", D);
    for (auto& x : aw2) x *= 0.5f;
    for (auto& x : ab2) x *= 0.5f;
    // distance bias MLP
    const std::string f = "mmg.self_attn_fc.";
    auto d0w = P.get(f + "0.weight", 32 * 4), d0b = P.get(f + "0.bias", 32);
    auto d2w = P.get(f + "2.weight", 32), d2b = P.get(f + "2.bias", 32);
    auto d3w = P.get(f + "3.weight", 32 * 32), d3b = P.get(f + "3.bias", 32);
    auto d5w = P.get(f + "5.weight", 32), d5b = P.get(f + "5.bias", 32);
    auto d6w = P.get(f + "6.weight", (size_t)h->H * 32), d6b = P.get(f + "6.bias", h->H);
    if (!P.missing.empty()) return fail(VLSAT_ESTATE, "missing weight: " + P.missing);

    UP(w1, h->pn_w1); UP(b1, h->pn_b1); UP(w2, h->pn_w2); UP(b2, h->pn_b2); UP(w3, h->pn_w3); UP(b3, h->pn_b3);
    UP(mw, h->mlp_w); UP(mb, h->mlp_b);
    UP(w1cat, h->re_w1cat); UP(b1cat, h->re_b1cat);
    UP(r3w2, h->re3_w2); UP(r3b2, h->re3_b2); UP(r3w3, h->re3_w3); UP(r3b3, h->re3_b3);
    UP(r2w2, h->re2_w2); UP(r2b2, h->re2_b2); UP(r2w3, h->re2_w3); UP(r2b3, h->re2_b3);
    UP(aw1, h->ad_w1); UP(ab1, h->ad_b1); UP(aw2, h->ad_w2h); UP(ab2, h->ad_b2h);
    float* t;
    UP(d0w, t); h->db.w0 = t; UP(d0b, t); h->db.b0 = t; UP(d2w, t); h->db.g2 = t; UP(d2b, t); h->db.be2 = t;
    UP(d3w, t); h->db.w3 = t; UP(d3b, t); h->db.b3 = t; UP(d5w, t); h->db.g5 = t; UP(d5b, t); h->db.be5 = t;
    UP(d6w, t); h->db.w6 = t; UP(d6b, t); h->db.b6 = t;

    h->self_attn.resize(L); h->cross_attn.resize(L); h->cross_rel.resize(L); h->gcn3.resize(L); h->gcn2.resize(L);
    for (int l = 0; l < L; ++l) {
        const std::string ls = std::to_string(l);
        RUN(prepare_attn(h, P, "mmg.self_attn." + ls, h->self_attn[l], true, 0.125f));
        RUN(prepare_attn(h, P, "mmg.cross_attn." + ls, h->cross_attn[l], false, 0.125f));
        RUN(prepare_attn(h, P, "mmg.cross_attn_rel." + ls, h->cross_rel[l], false, 1.0f));
        RUN(prepare_gcn(h, P, "mmg.gcn_3ds." + ls, h->gcn3[l]));
        RUN(prepare_gcn(h, P, "mmg.gcn_2ds." + ls, h->gcn2[l]));
        if (!P.missing.empty()) return fail(VLSAT_ESTATE, "missing weight: " + P.missing);
    }
    const int R = h->d.n_rel_class, K = h->d.n_obj_class;
    for (int i = 0; i < 2; ++i) {
        const std::string b = i == 0 ? "rel_predictor_3d" : "rel_predictor_2d";
        RelHeadW& r = i == 0 ? h->rel3 : h->rel2;
        auto f1 = P.get(b + ".fc1.weight", 512 * (size_t)D), f1b = P.get(b + ".fc1.bias", 512);
        auto f2 = P.get(b + ".fc2.weight", 256 * 512), f2b = P.get(b + ".fc2.bias", 256);
        auto f3 = P.get(b + ".fc3.weight", (size_t)R * 256), f3b = P.get(b + ".fc3.bias", R);
        if (!P.missing.empty()) return fail(VLSAT_ESTATE, "missing weight: " + P.missing);
        // MODEL.WITH_BN: BatchNorm1d(eval) after fc1 / fc2 (reference network_PointNet.py:320-337), recognised by
        // its keys in the checkpoint and folded into the layer in front of it
        auto fold = [&](const std::string& bn, std::vector<float>& wt, std::vector<float>& bs, int outs, int ins) -> int {
            if (!h->host.count(bn + ".weight")) return 0;
            auto g = P.get(bn + ".weight", outs), be = P.get(bn + ".bias", outs);
            auto mu = P.get(bn + ".running_mean", outs), var = P.get(bn + ".running_var", outs);
            if (!P.missing.empty()) return fail(VLSAT_ESTATE, "missing weight: " + P.missing);
            for (int o = 0; o < outs; ++o) {
                const double sc = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
                for (int k = 0; k < ins; ++k) wt[(size_t)o * ins + k] = (float)(wt[(size_t)o * ins + k] * sc);
                bs[o] = (float)(((double)bs[o] - mu[o]) * sc + be[o]);
            }
            return 0;
        };
        RUN(fold(b + ".bn1", f1, f1b, 512, D));
        RUN(fold(b + ".bn2", f2, f2b, 256, 512));
        UP(f1, r.w1); UP(f1b, r.b1); UP(f2, r.w2); UP(f2b, r.b2); UP(f3, r.w3); UP(f3b, r.b3);
    }
    const float es = std::exp(h->d.obj_logit_scale);
    for (int i = 0; i < 2; ++i) {
        const std::string b = i == 0 ? "obj_predictor_3d" : "obj_predictor_2d";
        auto w = P.get(b + ".weight", (size_t)K * D), bi = P.get(b + ".bias", K);
        if (!P.missing.empty()) return fail(VLSAT_ESTATE, "missing weight: " + P.missing);
        for (auto& x : bi) x *= es;     // exp(s) * (W x/|x| + b)
        if (i == 0) { UP(w, h->obj3_w); UP(bi, h->obj3_b); } else { UP(w, h->obj2_w); UP(bi, h->obj2_b); }
    }
    h->host.clear();
    h->finalized = true;
    return 0;
}

// -------------------------------------------------------------------------------------------
int vlsat_plan_create(vlsat_handle h, const int64_t* bid, const int64_t* edges, int64_t N, int64_t E, int32_t P,
                      vlsat_plan* out) {
    if (!h || !out || !bid || (!edges && E > 0)) return fail(VLSAT_EINVAL, "vlsat_plan_create: null argument");
    if (!h->finalized) return fail(VLSAT_ESTATE, "weights not finalised");
    if (N <= 0 || E < 0 || P <= 0) return fail(VLSAT_EINVAL, "N, P must be positive and E non-negative");
    if (N > (1 << 28) || E > (1ll << 30)) return fail(VLSAT_EINVAL, "graph too large for 32-bit indices");
    std::unique_ptr<vlsat_plan_s> p(new vlsat_plan_s());
    p->h = h; p->N = N; p->E = E; p->P = P;
    // ---- scenes: maximal runs of equal batch id (must not re-appear) ----
    std::vector<int32_t> node_scene(N);
    p->node_ptr.push_back(0);
    {
        std::map<int64_t, int> seen;
        for (int64_t i = 0; i < N; ++i) {
            if (i == 0 || bid[i] != bid[i - 1]) {
                if (seen.count(bid[i])) return fail(VLSAT_EINVAL, "batch_ids: nodes of a scene must be contiguous");
                seen[bid[i]] = 1;
                if (i) p->node_ptr.push_back((int32_t)i);
            }
            node_scene[i] = (int32_t)p->node_ptr.size() - 1;
        }
        p->node_ptr.push_back((int32_t)N);
    }
    p->S = (int)p->node_ptr.size() - 1;
    for (int s = 0; s < p->S; ++s) p->max_n = std::max(p->max_n, p->node_ptr[s + 1] - p->node_ptr[s]);
    // ---- edges: same-scene endpoints, grouped by scene in node order ----
    std::vector<int32_t> src(std::max<int64_t>(E, 1)), dst(std::max<int64_t>(E, 1));
    p->edge_ptr.assign(p->S + 1, 0);
    int cur = 0;
    bool sorted_by_src = true;
    for (int64_t e = 0; e < E; ++e) {
        const int64_t a = edges[e], b = edges[E + e];
        if (a < 0 || a >= N || b < 0 || b >= N) return fail(VLSAT_EINVAL, "edge index out of range");
        const int sa = node_scene[a];
        if (sa != node_scene[b]) return fail(VLSAT_EINVAL, "edge joins nodes of different scenes");
        if (sa < cur) return fail(VLSAT_EGRAPH, "edges are not grouped by scene in node order");
        while (cur < sa) p->edge_ptr[++cur] = e;
        src[e] = (int32_t)a; dst[e] = (int32_t)b;
        if (e && src[e] < src[e - 1]) sorted_by_src = false;
    }
    while (cur < p->S) p->edge_ptr[++cur] = E;
    // ---- CSR over sources (stable counting sort) ----
    std::vector<int32_t> rowptr(N + 1, 0), order(std::max<int64_t>(E, 1));
    for (int64_t e = 0; e < E; ++e) rowptr[src[e] + 1]++;
    for (int64_t i = 0; i < N; ++i) rowptr[i + 1] += rowptr[i];
    {
        std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
        for (int64_t e = 0; e < E; ++e) order[fill[src[e]]++] = (int32_t)e;
    }
    p->is_fc = sorted_by_src;
    for (int s = 0; s < p->S && p->is_fc; ++s) {
        const int64_t n = p->node_ptr[s + 1] - p->node_ptr[s];
        if (p->edge_ptr[s + 1] - p->edge_ptr[s] != n * (n - 1)) p->is_fc = 0;
    }
    // ---- flash tiles: scene-major, head, q-tile (consecutive ids share K/V -> same XCD) ----
    std::vector<int4> tiles;
    std::vector<int64_t> bias_ptr(p->S);
    int64_t bias_total = 0;
    if (h->edge_scope == 1 && E > 0) {       // reference multi-scene call: one attention over all edges
        for (int hh = 0; hh < h->H; ++hh)
            for (int64_t q0 = 0; q0 < E; q0 += FLASH_BQ) tiles.push_back(make_int4(0, (int)E, (int)q0, hh));
        p->flash_flops += 4.0 * (double)E * (double)E * h->D;
    }
    for (int s = 0; s < p->S; ++s) {
        const int64_t T = p->edge_ptr[s + 1] - p->edge_ptr[s];
        if (h->edge_scope == 0) {
            for (int hh = 0; hh < h->H; ++hh)
                for (int64_t q0 = 0; q0 < T; q0 += FLASH_BQ)
                    tiles.push_back(make_int4((int)p->edge_ptr[s], (int)T, (int)q0, hh));
            p->flash_flops += 4.0 * (double)T * (double)T * h->D;
        }
        const int64_t n = p->node_ptr[s + 1] - p->node_ptr[s];
        bias_ptr[s] = bias_total;
        bias_total += (int64_t)h->H * n * n;
    }
    // Few blocks (one scene alone: ceil(T/128)*8 ~ 100 for 256 CUs): cut every block's key range into `parts`
    // pieces so that about two rounds of 512 resident blocks exist; each piece keeps at least two key tiles.
    std::vector<int4> krange;
    if (!tiles.empty() && tiles.size() < 512 && h->fa_split) {
        int parts = (int)std::min<size_t>(16, 1024 / tiles.size());
        if (parts > 1) {
            std::vector<int4> split;
            for (const int4& t : tiles) {
                const int kt = (t.y + 31) / 32;
                const int ps = std::max(1, std::min(parts, kt / 2));          // parts actually used by this scene
                for (int q = 0; q < parts; ++q) {
                    split.push_back(t);
                    const int a = q < ps ? (int)((int64_t)kt * q / ps) : 0, b = q < ps ? (int)((int64_t)kt * (q + 1) / ps) : 0;
                    krange.push_back(make_int4(a, b, q, 0));
                }
            }
            tiles.swap(split);
            p->fa_parts = parts;
        }
    }
    p->n_tiles = (int)tiles.size();
    // ---- one device arena ----
    const size_t Ns = (size_t)N, Es = (size_t)std::max<int64_t>(E, 1);
    struct Item { void** dst; size_t bytes; };
    std::vector<Item> items;
    auto want = [&](auto** ptr, size_t count) { items.push_back({reinterpret_cast<void**>(ptr), count * sizeof(**ptr)}); };
    want(&p->d_src, Es); want(&p->d_dst, Es); want(&p->d_rowptr, Ns + 1); want(&p->d_order, Es);
    want(&p->d_scene_ptr, (size_t)p->S + 1); want(&p->d_bias_ptr, (size_t)p->S); want(&p->d_tiles, std::max<size_t>(tiles.size(), 1));
    want(&p->F, Ns * 768); want(&p->X3, Ns * 768); want(&p->X2, Ns * 768); want(&p->NP, Ns * 3328);
    want(&p->QKVn, Ns * 1536); want(&p->On, Ns * 512); want(&p->T256, Ns * 256); want(&p->T768, Ns * 768);
    want(&p->rs, Ns); want(&p->bias, (size_t)std::max<int64_t>(bias_total, 1));
    want(&p->H1, Es * 128); want(&p->H2, Es * 128); want(&p->E3, Es * 512); want(&p->E2, Es * 512);
    want(&p->Hbig, Es * 1024); want(&p->KP, Es * 512); want(&p->G, Es * 256);
    want(&p->Qe, Es * 512); want(&p->KVe, Es * 1024); want(&p->Oe, Es * 512);
    // launch-bound plans (every edge GEMM fits one round of the grid): second scratch set for the 2D twin stages
    p->dual = h->dual_stream && E > 0 && E <= 8192;
    if (p->dual) {
        want(&p->NP2, Ns * 3328); want(&p->Hbig2, Es * 1024); want(&p->KP2, Es * 512); want(&p->G2, Es * 256);
        want(&p->T768b, Ns * 768); want(&p->rs2, Ns); want(&p->H2b, Es * 128);
    }
    if (h->d.feature_transform) {
        // point rows R = N*P (objects) or E (relation encoders, P = 1), one phase at a time:
        //   rows [R,64] h1, [R,64], [R,128], [R,1024] STN convs (the last two double as conv2/conv3 of the main chain),
        //   [R,64] h1';  per object: 1024 + 512 + 256 + 4096
        const size_t R = std::max<size_t>(Ns * (size_t)P, Es), O = std::max(Ns, Es);
        p->stn_ws_floats = R * (64 + 64 + 128 + 1024 + 64) + O * (1024 + 512 + 256 + 4096);
        want(&p->stn_ws, p->stn_ws_floats);
    }
    if (p->fa_parts > 1) {
        want(&p->d_krange, krange.size());
        want(&p->fa_opart, (size_t)p->fa_parts * Es * 512);
        want(&p->fa_m, (size_t)p->fa_parts * Es * h->H); want(&p->fa_l, (size_t)p->fa_parts * Es * h->H);
    }
    size_t total = 0;
    for (auto& it : items) total += (it.bytes + 255) & ~size_t(255);
    {   // smallest pooled arena that fits (and is not absurdly larger), else a fresh allocation
        int best = -1;
        for (size_t i = 0; i < h->arena_pool.size(); ++i)
            if (h->arena_pool[i].second >= total && h->arena_pool[i].second <= 4 * total + (64u << 20) &&
                (best < 0 || h->arena_pool[i].second < h->arena_pool[best].second))
                best = (int)i;
        if (best >= 0) {
            p->arena = h->arena_pool[best].first;
            p->arena_bytes = h->arena_pool[best].second;
            h->arena_pool.erase(h->arena_pool.begin() + best);
        } else {
            VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p->arena), total));
            p->arena_bytes = total;
        }
    }
    size_t off = 0;
    for (auto& it : items) {
        *it.dst = p->arena + off;
        off += (it.bytes + 255) & ~size_t(255);
    }
    p->R1 = p->Hbig;                 // relation-head hidden layers re-use the nn_edge hidden buffer
    p->R2 = p->Hbig + Es * 512;
    p->prob = nullptr;
    p->ws_bytes = total;
    auto cp = [&](void* d, const void* s_, size_t b) { return hipMemcpy(d, s_, b, hipMemcpyHostToDevice); };
    hipError_t er = hipSuccess;
    if (E > 0) {
        if (er == hipSuccess) er = cp(p->d_src, src.data(), E * 4);
        if (er == hipSuccess) er = cp(p->d_dst, dst.data(), E * 4);
        if (er == hipSuccess) er = cp(p->d_order, order.data(), E * 4);
        if (er == hipSuccess && !tiles.empty()) er = cp(p->d_tiles, tiles.data(), tiles.size() * sizeof(int4));
        if (er == hipSuccess && !krange.empty()) er = cp(p->d_krange, krange.data(), krange.size() * sizeof(int4));
    }
    if (er == hipSuccess) er = cp(p->d_rowptr, rowptr.data(), (N + 1) * 4);
    if (er == hipSuccess) er = cp(p->d_scene_ptr, p->node_ptr.data(), (p->S + 1) * 4);
    if (er == hipSuccess) er = cp(p->d_bias_ptr, bias_ptr.data(), p->S * 8);
    if (er != hipSuccess) {
        hipFree(p->arena);
        p->arena = nullptr;
        return fail(VLSAT_EHIP, std::string("plan upload: ") + hipGetErrorString(er));
    }
    *out = p.release();
    return 0;
}

void vlsat_plan_destroy(vlsat_plan p) {
    if (!p) return;
    if (p->arena) {
        // the forward that used this workspace may still be in flight on some stream
        hipDeviceSynchronize();
        if (p->h && p->h->arena_pool.size() < 8) p->h->arena_pool.emplace_back(p->arena, p->arena_bytes);
        else hipFree(p->arena);
    }
    delete p;
}

int vlsat_plan_info(vlsat_plan p, int32_t* n_scenes, size_t* ws, int32_t* is_fc) {
    if (!p) return fail(VLSAT_EINVAL, "null plan");
    if (n_scenes) *n_scenes = p->S;
    if (ws) *ws = p->ws_bytes;
    if (is_fc) *is_fc = p->is_fc;
    return 0;
}

// debug: stop the forward after stage `stage` (see DESIGN.md "debug stages"); -1 = run all
int vlsat_debug_stop_after(vlsat_handle h, int32_t stage) {
    if (!h) return fail(VLSAT_EINVAL, "null handle");
    h->debug_stop = stage;
    return 0;
}
// debug: device pointer / shape of a named workspace buffer of a plan
int vlsat_debug_buffer(vlsat_plan p, const char* name, void** ptr, int64_t* rows, int32_t* cols, int32_t* ld) {
    if (!p || !name) return fail(VLSAT_EINVAL, "null argument");
    struct B { const char* n; float* p; int64_t r; int c, ld; };
    const B tab[] = {{"F", p->F, p->N, 768, 768},       {"X3", p->X3, p->N, 512, 768},     {"X2", p->X2, p->N, 512, 768},
                     {"AGG3", p->X3 + 512, p->N, 256, 768}, {"AGG2", p->X2 + 512, p->N, 256, 768},
                     {"E3", p->E3, p->E, 512, 512},     {"E2", p->E2, p->E, 512, 512},     {"G", p->G, p->E, 256, 256},
                     {"H1", p->H1, p->E, 128, 128},     {"KP", p->KP, p->E, 512, 512},     {"NP", p->NP, p->N, 3328, 3328},
                     {"Hbig", p->Hbig, p->E, 1024, 1024}, {"bias", p->bias, 1, 0, 0},      {"On", p->On, p->N, 512, 512},
                     {"Oe", p->Oe, p->E, 512, 512},     {"Qe", p->Qe, p->E, 512, 512},     {"KVe", p->KVe, p->E, 1024, 1024}};
    for (auto& b : tab)
        if (!std::strcmp(b.n, name)) {
            if (ptr) *ptr = b.p;
            if (rows) *rows = b.r;
            if (cols) *cols = b.c;
            if (ld) *ld = b.ld;
            return 0;
        }
    return fail(VLSAT_EINVAL, std::string("unknown buffer ") + name);
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
    return 0;
}

// debug: experimental GEMM variant for the full rounds of large-M fp32 launches (see vlsat.h)
int vlsat_debug_gemm_variant(int32_t variant) {
    if (variant < 0 || variant > 2) return fail(VLSAT_EINVAL, "gemm_variant: 0, 1 or 2");
    gemm_set_variant(variant);
    return 0;
}

// debug: synchronous strided copy of a named workspace buffer into dst (device, row pitch dst_ld floats)
int vlsat_debug_read(vlsat_plan p, const char* name, float* dst, int64_t dst_ld) {
    void* src = nullptr; int64_t rows = 0; int32_t cols = 0, ld = 0;
    int r = vlsat_debug_buffer(p, name, &src, &rows, &cols, &ld);
    if (r) return r;
    if (!dst || rows <= 0 || cols <= 0) return fail(VLSAT_EINVAL, "debug_read: nothing to copy");
    VLSAT_HIP_CHECK(hipDeviceSynchronize());
    VLSAT_HIP_CHECK(hipMemcpy2D(dst, (size_t)dst_ld * 4, src, (size_t)ld * 4, (size_t)cols * 4, (size_t)rows,
                                hipMemcpyDeviceToDevice));
    return 0;
}

// One encoder with MODEL.feature_transform: h1 [R,64] (row pitch ldh) -> T = STNkd(h1) per object (objects own P
// consecutive rows) -> h1' = h1 . T -> relu(conv2) -> relu(conv3) [R, n_out]; the caller takes the max over an
// object's rows (or uses the rows directly when P == 1).  Returns the conv3 output rows in *out_rows (pitch n_out).
namespace {
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

// -------------------------------------------------------------------------------------------
int vlsat_forward(vlsat_handle h, vlsat_plan p, const float* pts, const float* f2d, const float* desc,
                  float* obj3d, float* obj2d, float* rel3d, float* rel2d, void* stream) {
    if (!h || !p || !pts || !desc || !obj3d) return fail(VLSAT_EINVAL, "vlsat_forward: null argument");
    if (p->h != h) return fail(VLSAT_EINVAL, "plan belongs to a different handle");
    // 3D-only mode: both 2D outputs NULL -> the 2D branch (adapter, cross-attention, gcn_2ds, edge
    // cross-attention, 2D heads) is skipped.  Exact: the 3D branch never reads 2D tensors.
    const bool do2d = obj2d != nullptr || rel2d != nullptr;
    if (do2d && (!obj2d || !f2d || (p->E > 0 && !rel2d)))
        return fail(VLSAT_EINVAL, "vlsat_forward: 2D branch needs obj_2d_feats and both 2D outputs (or neither for 3D-only)");
    if (p->E > 0 && !rel3d) return fail(VLSAT_EINVAL, "vlsat_forward: null relation output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = (int)p->N, E = (int)p->E, D = h->D, L = h->d.n_layers, LDX = 768;
    const int stop = h->debug_stop;
    profile_close(h, s);                   // an interval left open by a failed forward must not span foreign work
#define STAGE(id) do { if (stop == (id)) { profile_close(h, s); return 0; } } while (0)

    // Two-stream mode (small plans only; not while profiling or stopping at a debug stage): `t` carries the 2D twin
    // of a stage while `s` carries the 3D one.  fork(): t waits for everything enqueued on s so far; join(): s waits
    // for t.  Every forward ends joined, so the caller only ever sees its own stream.
    const bool dual = p->dual && do2d && !h->prof && stop < 0;
    hipStream_t t = s;
    size_t ev_i = 0;
    if (dual) {
        if (!h->side) VLSAT_HIP_CHECK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        t = h->side;
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
    auto order = [&](hipStream_t first, hipStream_t then) -> int {       // `then` continues after `first`'s work so far
        if (!dual) return 0;
        hipEvent_t e;
        RUN(next_ev(&e));
        VLSAT_HIP_CHECK(hipEventRecord(e, first));
        VLSAT_HIP_CHECK(hipStreamWaitEvent(then, e, 0));
        return 0;
    };
    auto fork = [&]() { return order(s, t); };
    auto join = [&]() { return order(t, s); };
    const Scratch sc3 = scratch_of(p, 0), sc2 = scratch_of(p, dual ? 1 : 0);

    const bool ft = h->d.feature_transform != 0;
    if (!ft) {   // a-2 object encoder
        Scope sc(h, s, PC_POINTNET, 213376.0 * N * p->P);
        RUN(launch_pointnet(pts, N, p->P, h->d.dim_point, h->pn_w1, h->pn_b1, h->pn_w2, h->pn_b2, h->pn_w3, h->pn_b3, h->C_pt, p->F, s));
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
    RUN(fork());                                                            // t: 2D relation encoder + adapter
    if (!ft) {
        RUN(gemm(h, s, G(p->H1, 128, h->re3_w2, 64, sc3.H2, 128, E, 128, h->re3_b2, ACT_RELU)));
        RUN(gemm(h, s, G(sc3.H2, 128, h->re3_w3, 128, p->E3, D, E, D, h->re3_b3, ACT_RELU)));
        if (do2d) {
            RUN(gemm(h, t, G(p->H1 + 64, 128, h->re2_w2, 64, sc2.H2, 128, E, 128, h->re2_b2, ACT_RELU)));
            RUN(gemm(h, t, G(sc2.H2, 128, h->re2_w3, 128, p->E2, D, E, D, h->re2_b3, ACT_RELU)));
        }
    } else if (E > 0) {   // P = 1: one row per edge; both encoders share the scratch, so they run one after the other on s
        for (int br = 0; br < (do2d ? 2 : 1); ++br) {
            float* out = nullptr;
            RUN(stn_encoder(h, p, s, br ? h->stn_re2 : h->stn_re3, p->H1 + 64 * br, 128, (size_t)E, 1, br ? h->re2_w2 : h->re3_w2,
                            br ? h->re2_b2 : h->re3_b2, br ? h->re2_w3 : h->re3_w3, br ? h->re2_b3 : h->re3_b3, D, &out));
            VLSAT_HIP_CHECK(hipMemcpyAsync(br ? p->E2 : p->E3, out, (size_t)E * D * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    STAGE(3);
    // a-6 adapter -> X2[:, 0:512]
    if (do2d) {
        RUN(gemm(h, t, G(f2d, D, h->ad_w1, D, p->T256, 256, N, 256, h->ad_b1, ACT_RELU)));
        GemmArgs a = G(p->T256, 256, h->ad_w2h, 256, p->X2, LDX, N, D, h->ad_b2h);
        a.resid = f2d; a.ldr = D; a.resid_scale = 0.5f;
        RUN(gemm(h, t, a));
    }
    STAGE(4);
    {   // a-7 distance bias
        Scope sc(h, s, PC_MISC, 0);
        RUN(launch_dist_bias(desc, 11, p->d_scene_ptr, p->d_bias_ptr, p->S, p->max_n, h->H, h->db, p->bias, s));
    }
    STAGE(5);
    int e3_pending_relu = 0;
    for (int l = 0; l < L; ++l) {
        const int inter = (l < L - 1 || L == 1) ? 1 : 0;     // reference network_MMG.py:236
        const int base = 10 + 10 * l;
        RUN(attn_block(h, p, s, h->self_attn[l], p->X3, p->X3, true));                  // :217
        STAGE(base + 0);
        RUN(join());                                          // X2 / E2 of the previous stage are complete
        if (do2d) RUN(attn_block(h, p, s, h->cross_attn[l], p->X2, p->X3, false));      // :218
        STAGE(base + 1);
        RUN(fork());                                          // t: gcn_2ds + query projection; s: gcn_3ds + key/value projection
        RUN(gcn_block(h, p, s, h->gcn3[l], p->X3, p->E3, e3_pending_relu, inter, sc3)); // :224
        STAGE(base + 2);
        if (do2d) RUN(gcn_block(h, p, t, h->gcn2[l], p->X2, p->E2, 0, inter, sc2));     // :225
        STAGE(base + 3);
        if (do2d) {   // :231 edge cross-attention: q = 2D edges, k = v = 3D edges (pre-activation)
            const AttnW& w = h->cross_rel[l];
            RUN(gemm(h, t, G(p->E2, D, w.wq, D, p->Qe, D, E, D, w.bq)));
            RUN(gemm(h, s, G(p->E3, D, w.wkv, D, p->KVe, 2 * D, E, 2 * D, w.bkv)));
            RUN(join());
            {
                Scope sc(h, s, PC_FLASH, p->flash_flops);
                FlashSplit sp;
                sp.parts = p->fa_parts; sp.krange = p->d_krange; sp.o_part = p->fa_opart; sp.m_part = p->fa_m;
                sp.l_part = p->fa_l; sp.part_stride = (size_t)E * D; sp.rows = E; sp.heads = h->H;
                RUN(launch_flash_attn(p->Qe, D, p->KVe, p->KVe + D, 2 * D, p->Oe, D, p->d_tiles, p->n_tiles,
                                      0.125f * 1.4426950408889634f, s, &sp));
            }
            GemmArgs o = G(p->Oe, D, w.wo, D, p->E2, D, E, D, w.bo);
            o.resid = p->E2; o.ldr = D;
            RUN(gemm(h, s, o));
            Scope sc(h, s, PC_LAYERNORM, 0);
            RUN(launch_layernorm(p->E2, D, E, D, w.lng, w.lnb, inter, s));
        }
        e3_pending_relu = inter;
        STAGE(base + 4);
    }
    // a-15 relation heads, a-16 object heads
    RUN(fork());                                              // t: the 2D heads
    if (E > 0) {
        RUN(rel_head(h, p, s, h->rel3, p->E3, e3_pending_relu, rel3d, sc3));
        if (do2d) RUN(rel_head(h, p, t, h->rel2, p->E2, 0, rel2d, sc2));
    }
    RUN(obj_head(h, p, s, p->X3, h->obj3_w, h->obj3_b, obj3d, sc3));
    if (do2d) RUN(obj_head(h, p, t, p->X2, h->obj2_w, h->obj2_b, obj2d, sc2));
    RUN(join());
    profile_close(h, s);
#undef STAGE
    return 0;
}

// -------------------------------------------------------------------------------------------
int vlsat_set_gemm_precision(vlsat_handle h, int32_t mode) {
    if (!h) return fail(VLSAT_EINVAL, "null handle");
    if (mode != 0 && mode != 1 && mode != 3) return fail(VLSAT_EINVAL, "gemm precision must be 0 (fp32), 1 (bf16) or 3 (bf16x3)");
    h->prec = mode;
    return 0;
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
    VLSAT_HIP_CHECK(hipEventSynchronize(h->recs.back().b));
    for (auto& r : h->recs) {
        float ms = 0.f;
        VLSAT_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        h->acc_ms[r.cls] += ms;
        h->acc_n[r.cls] += r.kernels;
        h->acc_fl[r.cls] += r.flops;
    }
    h->recs.clear();
    h->ev_used = 0;
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

// -------------------------------------------------------------------------------------------
int vlsat_k_gemm(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc, int32_t M, int32_t N,
                 int32_t K, const float* bias, const float* rowscale, const float* resid, int32_t ldr, float resid_scale,
                 const float* g0, const int32_t* gi0, int32_t ldg0, const float* g1, const int32_t* gi1, int32_t ldg1,
                 int32_t relu_a, int32_t act, void* stream) {
    GemmArgs a;
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.bias = bias; a.rowscale = rowscale; a.resid = resid; a.ldr = ldr; a.resid_scale = resid_scale;
    a.g0 = g0; a.gi0 = gi0; a.ldg0 = ldg0; a.g1 = g1; a.gi1 = gi1; a.ldg1 = ldg1; a.relu_a = relu_a; a.act = act;
    return launch_gemm(a, static_cast<hipStream_t>(stream));
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

int vlsat_k_layernorm(float* x, int32_t ld, int32_t rows, int32_t dim, const float* gamma, const float* beta,
                      int32_t relu, void* stream) {
    return launch_layernorm(x, ld, rows, dim, gamma, beta, relu, static_cast<hipStream_t>(stream));
}

}  // extern "C"
