// Graph plan: everything about one (edge list, batch ids, point count) that does not depend on the tensors' values --
// scene partition, CSR over sources, attention tile tables, one device arena carved into every workspace buffer.
//
// No device-wide synchronisation anywhere on this path: the index tables are packed into ONE pinned host
// buffer and uploaded with ONE hipMemcpyAsync on the handle's copy stream; the first forward of the plan makes its
// stream wait for that upload's event.  A destroyed plan hands its arena to the pool together with the event of the
// last forward that touched it; whoever takes the arena next orders its upload behind that event, and arenas that
// fall out of the pool are only freed once the event has completed.
#include <algorithm>
#include <cstring>
#include <memory>

#include "engine.h"

using namespace vlsat;

namespace vlsat {

hipEvent_t take_event(vlsat_ctx* h) {
    if (!h->spare_ev.empty()) {
        hipEvent_t e = h->spare_ev.back();
        h->spare_ev.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
}
void give_event(vlsat_ctx* h, hipEvent_t e) {
    if (e) h->spare_ev.push_back(e);
}

// arenas that fell out of the pool: free those whose last forward has completed (hipFree may block on the device,
// so never call it on memory that is still in use)
static void sweep_trash(vlsat_ctx* h, bool force) {
    for (size_t i = 0; i < h->arena_trash.size();) {
        Arena& a = h->arena_trash[i];
        if (force || !a.last || hipEventQuery(a.last) == hipSuccess) {
            hipFree(a.p);
            give_event(h, a.last);
            h->arena_trash.erase(h->arena_trash.begin() + i);
        } else {
            ++i;
        }
    }
}

void release_plan_resources(vlsat_ctx* h) {
    sweep_trash(h, true);
    for (auto& a : h->arena_pool) { hipFree(a.p); if (a.last) hipEventDestroy(a.last); }
    h->arena_pool.clear();
    for (auto& s : h->staging) { hipHostFree(s.p); if (s.done) hipEventDestroy(s.done); }
    h->staging.clear();
    for (hipEvent_t e : h->spare_ev) hipEventDestroy(e);
    h->spare_ev.clear();
}

// a pinned buffer of at least `bytes` whose previous upload has completed
static int take_staging(vlsat_ctx* h, size_t bytes, Staging** out) {
    for (auto& s : h->staging)
        if (s.bytes >= bytes && hipEventQuery(s.done) == hipSuccess) { *out = &s; return 0; }
    if (h->staging.size() >= 16) {              // all busy (16 uploads in flight): wait for the oldest that fits, else the first
        Staging* pick = &h->staging[0];
        for (auto& s : h->staging) if (s.bytes >= bytes) { pick = &s; break; }
        VLSAT_HIP_CHECK(hipEventSynchronize(pick->done));
        if (pick->bytes < bytes) {
            hipHostFree(pick->p);
            pick->p = nullptr;
            pick->bytes = std::max<size_t>(bytes * 2, 1 << 16);
            VLSAT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&pick->p), pick->bytes, hipHostMallocDefault));
        }
        *out = pick;
        return 0;
    }
    Staging s;
    s.bytes = std::max<size_t>(bytes * 2, 1 << 16);
    VLSAT_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&s.p), s.bytes, hipHostMallocDefault));
    VLSAT_HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    h->staging.push_back(s);
    *out = &h->staging.back();
    return 0;
}

}  // namespace vlsat

extern "C" {

int vlsat_plan_create(vlsat_handle h, const int64_t* bid, const int64_t* edges, int64_t N, int64_t E, int32_t P,
                      vlsat_plan* out) {
    if (!h || !out || !bid || (!edges && E > 0)) return fail(VLSAT_EINVAL, "vlsat_plan_create: null argument");
    if (!h->finalized) return fail(VLSAT_ESTATE, "weights not finalised");
    if (N <= 0 || E < 0 || P <= 0) return fail(VLSAT_EINVAL, "N, P must be positive and E non-negative");
    if (N > (1 << 28) || E > (1ll << 30)) return fail(VLSAT_EINVAL, "graph too large for 32-bit indices");
    std::unique_ptr<vlsat_plan_s> p(new vlsat_plan_s());
    p->h = h; p->N = N; p->E = E; p->P = P;
    const int H = h->H, D = h->D;
    const size_t LDX = (size_t)ldx_of(h), NPC = (size_t)npc_of(h), A = (size_t)h->A;
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
    const size_t Es = (size_t)std::max<int64_t>(E, 1), Ns = (size_t)N;
    std::vector<int32_t> src(Es), dst(Es);
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
    std::vector<int32_t> rowptr(N + 1, 0), order(Es);
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
        for (int hh = 0; hh < H; ++hh)
            for (int64_t q0 = 0; q0 < E; q0 += FLASH_BQ) tiles.push_back(make_int4(0, (int)E, (int)q0, hh));
        p->flash_flops += 4.0 * (double)E * (double)E * D;
    }
    for (int s = 0; s < p->S; ++s) {
        const int64_t T = p->edge_ptr[s + 1] - p->edge_ptr[s];
        if (h->edge_scope == 0) {
            for (int hh = 0; hh < H; ++hh)
                for (int64_t q0 = 0; q0 < T; q0 += FLASH_BQ)
                    tiles.push_back(make_int4((int)p->edge_ptr[s], (int)T, (int)q0, hh));
            p->flash_flops += 4.0 * (double)T * (double)T * D;
        }
        const int64_t n = p->node_ptr[s + 1] - p->node_ptr[s];
        bias_ptr[s] = bias_total;
        bias_total += (int64_t)H * n * n;
    }
    // Many blocks (several rounds of the resident slots): the kernels map block b to tile xcd_remap(b) -- XCD b % 8 walks a
    // contiguous range of tile ids in order -- and a scene's last query tile is usually mostly empty (1560 = 12 * 128 + 24:
    // one wave of four has work).  Those light tiles go to the END of every XCD's range, so that the last, partly filled
    // round of blocks is made of light work instead of ending on full tiles next to idle CUs.
    if (tiles.size() >= 2048) {
        std::vector<int4> full, part;
        for (const int4& t : tiles) (t.z + FLASH_BQ <= t.y ? full : part).push_back(t);
        if (!part.empty() && !full.empty()) {
            const size_t n = tiles.size(), q = n / 8, r = n % 8;
            std::vector<int4> out;
            out.reserve(n);
            size_t fi = 0, pi = 0;
            for (size_t x = 0; x < 8; ++x) {
                const size_t cnt = q + (x < r ? 1 : 0);
                size_t np = part.size() * (x + 1) / 8 - part.size() * x / 8;          // this XCD's share of the light tiles
                np = std::min(np, cnt);
                size_t nf = std::min(cnt - np, full.size() - fi);
                np = cnt - nf;                                                          // (whatever the full list cannot cover)
                for (size_t i = 0; i < nf; ++i) out.push_back(full[fi++]);
                for (size_t i = 0; i < np && pi < part.size(); ++i) out.push_back(part[pi++]);
            }
            while (fi < full.size()) out.push_back(full[fi++]);                         // (rounding leftovers, if any)
            while (pi < part.size()) out.push_back(part[pi++]);
            if (out.size() == n) tiles.swap(out);
        }
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
    // Scenes of thousands of edges (cfg 5: one of 39 800): 256 queries per block share every K / V tile -- half the L2 -> LDS bytes
    // per query, and the partly filled last tile is < 1/16 of a scene.  Built only when EVERY scene is that large (one table, one
    // block size per launch); the forward uses it for half rows at head dim 64 (engine_forward.hip).
    std::vector<int4> tiles_big;
    if (h->edge_scope == 0 && p->fa_parts <= 1 && E > 0 && E * (int64_t)(2 * D) * 4 < (int64_t)1 << 32) {
        int64_t min_t = INT64_MAX;
        for (int s = 0; s < p->S; ++s) { const int64_t T = p->edge_ptr[s + 1] - p->edge_ptr[s]; if (T > 0) min_t = std::min(min_t, T); }
        if (min_t >= h->flash_bq_big_min && min_t != INT64_MAX)
            for (int s = 0; s < p->S; ++s) {
                const int64_t T = p->edge_ptr[s + 1] - p->edge_ptr[s];
                for (int hh = 0; hh < H; ++hh)
                    for (int64_t q0 = 0; q0 < T; q0 += FLASH_BQ_BIG) tiles_big.push_back(make_int4((int)p->edge_ptr[s], (int)T, (int)q0, hh));
            }
        if (tiles_big.size() < 1024) tiles_big.clear();          // (two rounds of the 512 resident blocks, as for the key split above)
    }
    p->n_tiles_big = (int)tiles_big.size();
    // ---- one device arena; the index tables come first, in the order they are packed into the staging buffer ----
    struct Item { void** dst; size_t bytes; const void* host; };
    std::vector<Item> items;
    auto want = [&](auto** ptr, size_t count, const void* host = nullptr) {
        items.push_back({reinterpret_cast<void**>(ptr), count * sizeof(**ptr), host});
    };
    want(&p->d_src, Es, src.data()); want(&p->d_dst, Es, dst.data()); want(&p->d_order, Es, order.data());
    want(&p->d_rowptr, Ns + 1, rowptr.data()); want(&p->d_scene_ptr, (size_t)p->S + 1, p->node_ptr.data());
    want(&p->d_bias_ptr, (size_t)p->S, bias_ptr.data());
    std::vector<int32_t> edge_ptr32(p->edge_ptr.begin(), p->edge_ptr.end());
    if (h->edge_scope == 1) { edge_ptr32.assign((size_t)p->S + 1, (int32_t)E); edge_ptr32[0] = 0; }   // one range: the whole batch
    for (int sc = 0; sc < p->S; ++sc) p->max_e = std::max<int>(p->max_e, (int)(p->edge_ptr[sc + 1] - p->edge_ptr[sc]));
    want(&p->d_edge_ptr32, (size_t)p->S + 1, edge_ptr32.data());
    want(&p->d_tiles, std::max<size_t>(tiles.size(), 1), tiles.empty() ? nullptr : tiles.data());
    if (p->fa_parts > 1) want(&p->d_krange, krange.size(), krange.data());
    if (!tiles_big.empty()) want(&p->d_tiles_big, tiles_big.size(), tiles_big.data());
    const size_t n_index_items = items.size();
    want(&p->F, Ns * 768); want(&p->X3, Ns * LDX); want(&p->X2, Ns * LDX); want(&p->NP, Ns * NPC);
    want(&p->QKVn, Ns * 1536); want(&p->On, Ns * 512); want(&p->T256, Ns * 256); want(&p->T768, Ns * LDX);
    want(&p->rs, Ns); want(&p->bias, (size_t)std::max<int64_t>(bias_total, 1));
    want(&p->H1, Es * 128); want(&p->H2, Es * 128); want(&p->E3, Es * 512); want(&p->E2, Es * 512);
    want(&p->Hbig, Es * 1024); want(&p->KP, Es * 512); want(&p->G, Es * A);
    want(&p->Qe, Es * 512); want(&p->KVe, Es * 1024); want(&p->Oe, Es * 512);
    want(&p->Q2n, Ns * 512); want(&p->On2, Ns * 512);
    {   // evaluation scratch (vlsat_process_val_counts): 0.6 KB per edge, 2.6 KB per node
        const size_t C = (size_t)h->d.n_obj_class, R = (size_t)h->d.n_rel_class;
        want(&p->ev_f, Ns * C * 4 + Es * R * 2 + Ns * C);      // (+ [N, <= C] sorted probabilities of the triplet ranking)
        want(&p->ev_i, Ns * 2 + Es * R * 4 + Es * 2);
    }
    // launch-bound plans (every edge GEMM fits one round of the grid): second scratch set for the 2D twin stages
    p->dual = h->dual_stream && E > 0 && (h->dual_stream > 1 || E <= 8192);      // (dual_stream = 2: every plan)
    if (p->dual) {                         // ... unless the second scratch set would take the plan past the budget
        size_t base = 0;
        for (auto& it : items) base += it.bytes;
        const size_t extra = (Ns * (size_t)(NPC + LDX + 1 + 1024 * (size_t)h->d.n_layers) + Es * (size_t)(1024 + 512 + A + 128 + 1024)) * sizeof(float);
        if (base + extra > DUAL_WS_BUDGET) p->dual = false;
    }
    if (p->dual) {
        want(&p->NP2, Ns * NPC); want(&p->Hbig2, Es * 1024); want(&p->KP2, Es * 512); want(&p->G2, Es * A);
        want(&p->T768b, Ns * LDX); want(&p->rs2, Ns); want(&p->H2b, Es * 128);
        want(&p->KVe2, Es * 1024);
        p->kvx_slots = std::max(1, (int)h->d.n_layers);
    }
    want(&p->KVx, Ns * 1024 * (size_t)p->kvx_slots);
    if (h->d.feature_transform) {
        // point rows R = N*P (objects) or E (relation encoders, P = 1), one phase at a time:
        //   rows [R,64] h1, [R,64], [R,128], [R,1024] STN convs (the last two double as conv2/conv3 of the main chain),
        //   [R,64] h1';  per object: 1024 + 512 + 256 + 4096
        const size_t R = std::max<size_t>(Ns * (size_t)P, Es), O = std::max(Ns, Es);
        p->stn_ws_floats = R * (64 + 64 + 128 + 1024 + 64) + O * (1024 + 512 + 256 + 4096);
        want(&p->stn_ws, p->stn_ws_floats);
    }
    if (p->fa_parts > 1) {
        want(&p->fa_opart, (size_t)p->fa_parts * Es * 512);
        want(&p->fa_m, (size_t)p->fa_parts * Es * H); want(&p->fa_l, (size_t)p->fa_parts * Es * H);
    }
    auto pad = [](size_t b) { return (b + 255) & ~size_t(255); };
    size_t total = 0, index_bytes = 0;
    for (size_t i = 0; i < items.size(); ++i) {
        total += pad(items[i].bytes);
        if (i + 1 == n_index_items) index_bytes = total;
    }
    sweep_trash(h, false);
    hipEvent_t prev_use = nullptr;
    {   // smallest pooled arena that fits (and is not absurdly larger), else a fresh allocation of the next SIZE CLASS
        // (2^k or 1.5 * 2^k bytes): an evaluation loop sees a new graph size almost every scene, and with exact sizes
        // nearly every plan would allocate and nearly every evicted one would end in hipFree (which waits for the device:
        // profiles/r02_hip_api_trace.txt had 48 of them in 80 forwards before the classes)
        int best = -1;
        for (size_t i = 0; i < h->arena_pool.size(); ++i)
            if (h->arena_pool[i].bytes >= total && h->arena_pool[i].bytes <= 4 * total + (64u << 20) &&
                (best < 0 || h->arena_pool[i].bytes < h->arena_pool[best].bytes))
                best = (int)i;
        size_t cls = size_t(1) << 20;
        while (cls < total) cls = (cls & (cls - 1)) ? (cls / 3) * 4 : cls + cls / 2;      // 1, 1.5, 2, 3, 4, 6, ... MiB
        if (best >= 0) {
            p->arena = h->arena_pool[best].p;
            p->arena_bytes = h->arena_pool[best].bytes;
            prev_use = h->arena_pool[best].last;
            h->arena_pool.erase(h->arena_pool.begin() + best);
        } else {
            VLSAT_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p->arena), cls));
            p->arena_bytes = cls;
        }
    }
    size_t off = 0;
    for (auto& it : items) {
        *it.dst = p->arena + off;
        off += pad(it.bytes);
    }
    p->R1 = p->Hbig;                 // relation-head hidden layers re-use the nn_edge hidden buffer
    p->R2 = p->Hbig + Es * 512;
    p->prob = nullptr;
    p->ws_bytes = total;
    // ---- pack + one asynchronous upload ----
    auto give_up = [&](int code, const std::string& msg) {
        h->arena_pool.push_back({p->arena, p->arena_bytes, prev_use});
        p->arena = nullptr;
        return fail(code, msg);
    };
    if (!h->copy && hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking) != hipSuccess)
        return give_up(VLSAT_EHIP, "plan: cannot create the copy stream");
    Staging* st = nullptr;
    if (take_staging(h, index_bytes, &st)) return give_up(VLSAT_EHIP, std::string("plan staging: ") + vlsat_last_error());
    off = 0;
    for (size_t i = 0; i < n_index_items; ++i) {
        if (items[i].host) std::memcpy(st->p + off, items[i].host, items[i].bytes);
        off += pad(items[i].bytes);
    }
    p->uploaded = take_event(h);
    p->last_use = take_event(h);
    hipError_t er = (p->uploaded && p->last_use) ? hipSuccess : hipErrorOutOfMemory;
    if (er == hipSuccess && prev_use) er = hipStreamWaitEvent(h->copy, prev_use, 0);     // the arena's previous owner is done
    if (er == hipSuccess) er = hipMemcpyAsync(p->arena, st->p, index_bytes, hipMemcpyHostToDevice, h->copy);
    if (er == hipSuccess) er = hipEventRecord(st->done, h->copy);
    if (er == hipSuccess) er = hipEventRecord(p->uploaded, h->copy);
    if (er != hipSuccess) {
        give_event(h, p->uploaded); give_event(h, p->last_use);
        return give_up(VLSAT_EHIP, std::string("plan upload: ") + hipGetErrorString(er));
    }
    give_event(h, prev_use);       // (stream-ordered: the wait above has been enqueued; the handle may re-record it later)
    p->upload_pending = true;
    *out = p.release();
    return 0;
}

void vlsat_plan_destroy(vlsat_plan p) {
    if (!p) return;
    vlsat_ctx* h = p->h;
    if (p->arena) {
        // The forward that used this workspace may still be in flight: the arena keeps the event of that forward (or
        // of the upload, if the plan never ran) and whoever takes it next waits for it ON THE DEVICE.  No host wait.
        Arena a{p->arena, p->arena_bytes, p->used ? p->last_use : p->uploaded};
        give_event(h, p->used ? p->uploaded : p->last_use);
        // the pool is bounded by bytes (8 GiB of 288) and count; beyond that the largest pooled arena goes
        h->arena_pool.push_back(a);
        size_t pooled = 0;
        for (auto& x : h->arena_pool) pooled += x.bytes;
        while (h->arena_pool.size() > 1 && (pooled > (size_t(8) << 30) || h->arena_pool.size() > 64)) {
            size_t big = 0;
            for (size_t i = 1; i < h->arena_pool.size(); ++i) if (h->arena_pool[i].bytes > h->arena_pool[big].bytes) big = i;
            pooled -= h->arena_pool[big].bytes;
            h->arena_trash.push_back(h->arena_pool[big]);
            h->arena_pool.erase(h->arena_pool.begin() + big);
        }
        sweep_trash(h, false);
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

// Does a DEVICE copy of the graph equal what this plan was built from?  *mismatches (device int32, zeroed by the caller) gets the
// number of edge columns of edges_dev [2,E] (int64) that differ from the plan's (from, to) tables plus the nodes at which
// batch_ids_dev (int64 [N], may be NULL) starts a run where the plan has no scene boundary or vice versa.  Asynchronous on
// `stream`.  For callers that name a graph by a key instead of handing the edge list over the host (VLSATModel's `fc_sizes`
// hint with device tensors): one call per new key makes the key's claim checked instead of trusted -- a wrongly ordered edge
// list would otherwise attribute every rel_cls row to the wrong edge (reference edge order: dataset_3dssg.py:264-266).
int vlsat_plan_check_graph(vlsat_plan p, const int64_t* edges_dev, const int64_t* batch_ids_dev, int32_t* mismatches, void* stream) {
    if (!p || !mismatches || (p->E > 0 && !edges_dev)) return fail(VLSAT_EINVAL, "vlsat_plan_check_graph: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p->upload_pending) {               // the plan's tables travel on the handle's copy stream
        VLSAT_HIP_CHECK(hipStreamWaitEvent(s, p->uploaded, 0));
        if (hipEventQuery(p->uploaded) == hipSuccess) p->upload_pending = false;
    }
    RUN(launch_check_graph(edges_dev, p->E, p->d_src, p->d_dst, batch_ids_dev, p->N, p->d_scene_ptr, p->S, mismatches, s));
    p->used = true;                        // (the arena's next owner must order behind this read)
    VLSAT_HIP_CHECK(hipEventRecord(p->last_use, s));
    return 0;
}

// debug: device pointer / shape of a named workspace buffer of a plan
int vlsat_debug_buffer(vlsat_plan p, const char* name, void** ptr, int64_t* rows, int32_t* cols, int32_t* ld) {
    if (!p || !name) return fail(VLSAT_EINVAL, "null argument");
    struct B { const char* n; float* p; int64_t r; int c, ld; };
    const int LDX = ldx_of(p->h), NPC = npc_of(p->h), A = p->h->A;
    const B tab[] = {{"F", p->F, p->N, 768, 768},       {"X3", p->X3, p->N, 512, LDX},     {"X2", p->X2, p->N, 512, LDX},
                     {"AGG3", p->X3 + 512, p->N, A, LDX}, {"AGG2", p->X2 + 512, p->N, A, LDX},
                     {"E3", p->E3, p->E, 512, 512},     {"E2", p->E2, p->E, 512, 512},     {"G", p->G, p->E, A, A},
                     {"H1", p->H1, p->E, 128, 128},     {"KP", p->KP, p->E, 512, 512},     {"NP", p->NP, p->N, NPC, NPC},
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

}  // extern "C"
