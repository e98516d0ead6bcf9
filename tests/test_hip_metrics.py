"""GPU eval-ranking kernels (csrc/eval_ranks.hip via the C ABI) against the reference goldens and
the metrics oracle.  Integer outputs: bit-exact given the same fp32 probabilities."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vlsat_amd  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_ranks_equal_reference_goldens(golden_dir, case):
    _need_gpu()
    from vlsat_amd import metrics as M
    z = np.load(os.path.join(golden_dir, "metrics_small.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    obj3 = None
    for suffix in ("", "_2d"):
        logits = g("obj_logits" + suffix)
        probs = F.softmax(logits, dim=-1)            # the reference's CPU softmax -> bit-exact triple scores
        r = M.eval_ranks(logits.to(DEV), g("rel" + suffix).to(DEV), g("gt_cls").to(DEV), g("gt_rel").to(DEV),
                         g("edges").to(DEV), obj_probs=probs.to(DEV))
        torch.cuda.synchronize()
        assert np.array_equal(r["top_k_obj"].cpu().numpy(), z[f"{case}.top_k_obj{suffix}"])
        assert np.array_equal(r["top_k_rel"].cpu().numpy(), z[f"{case}.top_k_rel{suffix}"])
        assert np.array_equal(r["top_k_triplet"].cpu().numpy(), z[f"{case}.top_k_triplet{suffix}"])
        if suffix == "":
            obj3 = r["top_k_obj"]
            cm = M.cls_matrix(g("gt_cls").to(DEV), g("gt_rel").to(DEV), g("edges").to(DEV), obj3)
            assert np.array_equal(cm.cpu().numpy(), z[f"{case}.cls_matrix"])
            s = M.summarize(r["top_k_obj"].cpu(), r["top_k_rel"].cpu(), r["top_k_triplet"].cpu(), cm.cpu())
            assert np.allclose([s["mean_recall@50"], s["mean_recall@100"]], z[f"{case}.mean_recall"])
        # with the GPU softmax the scores differ in the last bits at most: ranks agree (ties are measure-zero)
        r2 = M.eval_ranks(logits.to(DEV), g("rel" + suffix).to(DEV), g("gt_cls").to(DEV), g("gt_rel").to(DEV),
                          g("edges").to(DEV))
        assert float((r2["obj_probs"].cpu() - probs).abs().max()) < 1e-6
        agree = (r2["top_k_triplet"].cpu().numpy() == z[f"{case}.top_k_triplet{suffix}"]).mean()
        assert agree >= 0.98, agree


@pytest.mark.parametrize("case", ["a", "b"])
def test_single_label_ranks_equal_reference_goldens(golden_dir, case):
    """multi_rel_outputs=False (SURVEY 8a switch table): [E] label targets with 0 = none, log-probabilities in."""
    _need_gpu()
    from vlsat_amd import metrics as M
    z = np.load(os.path.join(golden_dir, "metrics_single_label.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    logits, rel = g("obj_logits"), g("rel")
    r = M.eval_ranks(logits.to(DEV), rel.to(DEV), g("gt_cls").to(DEV), g("gt_rel").to(DEV), g("edges").to(DEV),
                     obj_probs=F.softmax(logits, dim=-1).to(DEV), multi_rel_outputs=False, rel_exp=rel.exp().to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(r["top_k_obj"].cpu().numpy(), z[f"{case}.top_k_obj"])
    assert np.array_equal(r["top_k_rel"].cpu().numpy(), z[f"{case}.top_k_rel"])
    assert np.array_equal(r["top_k_triplet"].cpu().numpy(), z[f"{case}.top_k_triplet"])
    hot = M.multihot_targets(g("gt_rel").to(DEV), rel.shape[1])
    cm = M.cls_matrix(g("gt_cls").to(DEV), hot, g("edges").to(DEV), r["top_k_obj"])
    assert np.array_equal(cm.cpu().numpy(), z[f"{case}.cls_matrix"])
    # device-side exp(): the same ranks up to last-bit ties
    r2 = M.eval_ranks(logits.to(DEV), rel.to(DEV), g("gt_cls").to(DEV), g("gt_rel").to(DEV), g("edges").to(DEV),
                      multi_rel_outputs=False)
    assert (r2["top_k_triplet"].cpu().numpy() == z[f"{case}.top_k_triplet"]).mean() >= 0.98


def test_ranks_equal_oracle_random_graph():
    _need_gpu()
    from vlsat_amd import metrics as M
    from oracle import metrics_oracle as MO
    g = torch.Generator().manual_seed(42)
    n, e, C, R = 60, 300, 160, 26
    logits = torch.randn(n, C, generator=g) * 4
    gt = torch.randint(0, C, (n,), generator=g)
    logits[torch.arange(0, n, 3), gt[::3]] += 10
    edges = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)], 1)
    rel = torch.sigmoid(torch.randn(e, R, generator=g) * 3)
    gt_rel = (torch.rand(e, R, generator=g) < 0.07).long()
    gt_rel[5] = 1                                     # an edge with all 26 labels (> 4 thresholds path)
    probs = F.softmax(logits, dim=-1)
    r = M.eval_ranks(logits.to(DEV), rel.to(DEV), gt.to(DEV), gt_rel.to(DEV), edges.to(DEV), obj_probs=probs.to(DEV))
    torch.cuda.synchronize()
    obj = MO.topk_object(logits, gt, 11)
    assert np.array_equal(r["top_k_obj"].cpu().numpy(), obj)
    assert np.array_equal(r["top_k_rel"].cpu().numpy(), MO.topk_predicate(rel, gt_rel, 6))
    tri, cm = MO.triplet_topk(logits, rel, gt, gt_rel, edges, 101, obj, obj_probs=probs)
    assert np.array_equal(r["top_k_triplet"].cpu().numpy(), tri)
    assert np.array_equal(M.cls_matrix(gt.to(DEV), gt_rel.to(DEV), edges.to(DEV), r["top_k_obj"]).cpu().numpy(), cm)


def test_process_val_tuple_matches_oracle_pipeline():
    """End to end: forward + ranking through the process_val mirror, against oracle forward +
    metrics oracle on the same scene."""
    _need_gpu()
    from vlsat_amd import VLSATConfig, synth, metrics as M
    from vlsat_amd.model import VLSATModel
    from oracle import vlsat_oracle as O, metrics_oracle as MO
    cfg = VLSATConfig(N_LAYERS=2)
    w = synth.make_weights(cfg)
    b = synth.make_batch(1, 8, 64, seed0=8000)
    g = torch.Generator().manual_seed(1)
    n, e = 8, b["edge_indices"].shape[1]
    gt_cls = torch.randint(0, 160, (n,), generator=g)
    gt_rel = (torch.rand(e, 26, generator=g) < 0.06).long()
    edges = torch.from_numpy(b["edge_indices"]).t().contiguous()             # [E,2] as the loader yields
    model = VLSATModel(cfg, DEV).load_state(w).eval()
    d = {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}
    out = M.process_val(model, d["obj_points"], d["obj_2d_feats"], gt_cls.to(DEV), d["descriptor"], gt_rel.to(DEV),
                        edges.to(DEV), d["batch_ids"])
    c = {k: torch.from_numpy(v) for k, v in b.items()}
    ref = O.forward(O.to_torch(w), cfg, c["obj_points"], c["obj_2d_feats"], c["edge_indices"], c["descriptor"], c["batch_ids"])
    obj3 = MO.topk_object(ref[0], gt_cls, 11)
    assert np.array_equal(out[0], obj3) and np.array_equal(out[1], MO.topk_object(ref[1], gt_cls, 11))
    assert np.array_equal(out[2], MO.topk_predicate(ref[2], gt_rel, 6))
    assert np.array_equal(out[3], MO.topk_predicate(ref[3], gt_rel, 6))
    tri, cm = MO.triplet_topk(ref[0], ref[2], gt_cls, gt_rel, edges, 101, obj3)
    assert (out[4] == tri).mean() >= 0.97 and np.array_equal(out[6], cm)     # logits differ by ~4e-6: near-ties may flip
    assert out[7].shape == (int(gt_rel.sum()), 160) and out[9].shape == (int(gt_rel.sum()), 26)


def test_process_val_single_label_setting():
    """multi_rel_outputs=False end to end: log_softmax head over 27 classes, [E] label targets, ranks vs the oracles."""
    _need_gpu()
    from vlsat_amd import VLSATConfig, synth, metrics as M
    from vlsat_amd.model import VLSATModel
    from oracle import vlsat_oracle as O, metrics_oracle as MO
    cfg = VLSATConfig(**synth.SWITCH_CASES["switch_single_rel"])
    w = synth.make_weights(cfg)
    b = synth.make_batch(1, 7, 48, seed0=8100)
    g = torch.Generator().manual_seed(2)
    n, e = 7, b["edge_indices"].shape[1]
    gt_cls = torch.randint(0, 160, (n,), generator=g)
    gt_rel = torch.randint(0, 27, (e,), generator=g)
    gt_rel[torch.rand(e, generator=g) < 0.5] = 0
    edges = torch.from_numpy(b["edge_indices"]).t().contiguous()
    model = VLSATModel(cfg, DEV).load_state(w).eval()
    d = {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}
    out = M.process_val(model, d["obj_points"], d["obj_2d_feats"], gt_cls.to(DEV), d["descriptor"], gt_rel.to(DEV),
                        edges.to(DEV), d["batch_ids"])
    c = {k: torch.from_numpy(v) for k, v in b.items()}
    ref = O.forward(O.to_torch(w), cfg, c["obj_points"], c["obj_2d_feats"], c["edge_indices"], c["descriptor"], c["batch_ids"])
    assert float((ref[2].exp().sum(1) - 1).abs().max()) < 1e-5                  # the head really is a log_softmax
    obj3 = MO.topk_object(ref[0], gt_cls, 11)
    rel3, tri3, cm = MO.single_label_ranks(ref[0], ref[2], gt_cls, gt_rel, edges, 6, 101, obj3)
    assert np.array_equal(out[0], obj3)
    assert (out[2] == rel3).mean() >= 0.97 and (out[4] == tri3).mean() >= 0.97   # ~1e-6 output differences: near-ties may flip
    assert np.array_equal(out[6], cm)
    assert out[9].shape == (int((gt_rel > 0).sum()), 27) and float(out[9].sum(1).sub(1).abs().max()) < 1e-4


def test_sharded_validation_loop_single_rank():
    """evaluate.validation (forward + ranking + counts + the final all-reduce, here world=1) against the
    same pipeline built from the two oracles."""
    _need_gpu()
    from vlsat_amd import VLSATConfig, synth, evaluate as EV
    from vlsat_amd.model import VLSATModel
    from oracle import vlsat_oracle as O, metrics_oracle as MO
    cfg = VLSATConfig(N_LAYERS=1)
    w = synth.make_weights(cfg)
    model = VLSATModel(cfg, DEV).load_state(w).eval()
    g = torch.Generator().manual_seed(3)
    batches, vec = [], np.zeros(len(EV.fields()))
    for s in range(3):
        b = synth.collate([synth.make_scene(5 + s, 32, 9000 + 2 * s), synth.make_scene(4, 32, 9001 + 2 * s)])
        n, e = b["obj_points"].shape[0], b["edge_indices"].shape[1]
        gt_cls = torch.randint(0, 160, (n,), generator=g)
        gt_rel = (torch.rand(e, 26, generator=g) < 0.08).long()
        edges = torch.from_numpy(b["edge_indices"]).t().contiguous()
        item = {k: torch.from_numpy(v).to(DEV) for k, v in b.items() if k != "edge_indices"}
        item.update(gt_class=gt_cls.to(DEV), gt_rel_cls=gt_rel.to(DEV), edge_indices=edges.to(DEV))
        batches.append(item)
        c = {k: torch.from_numpy(v) for k, v in b.items()}
        ref = O.forward(O.to_torch(w), cfg, c["obj_points"], c["obj_2d_feats"], c["edge_indices"], c["descriptor"], c["batch_ids"])
        o3, o2 = MO.topk_object(ref[0], gt_cls, 11), MO.topk_object(ref[1], gt_cls, 11)
        t3, cm = MO.triplet_topk(ref[0], ref[2], gt_cls, gt_rel, edges, 101, o3)
        t2, _ = MO.triplet_topk(ref[1], ref[3], gt_cls, gt_rel, edges, 101, o3)
        EV.accumulate(vec, dict(top_k_obj=o3, top_k_obj_2d=o2, top_k_rel=MO.topk_predicate(ref[2], gt_rel, 6),
                                top_k_rel_2d=MO.topk_predicate(ref[3], gt_rel, 6), top_k_triplet=t3, top_k_triplet_2d=t2),
                      cm, 2)
    got = EV.validation(model, batches, device=DEV)
    want = EV.summarize(vec)
    assert got["scenes"] == 6
    for k in want:
        assert abs(got[k] - want[k]) <= 2.0, (k, got[k], want[k])      # percentages; near-tie flips move single counts
    exact = [k for k in want if k.startswith(("obj_acc", "rel_acc"))]
    assert all(got[k] == want[k] for k in exact)


def test_model_object_surface_like_the_reference(tmp_path, golden_dir):
    """What MMGNet touches on the model object: load(best) from the per-module checkpoint directory,
    eval(), process_val(...), iteration / eva_res (reference src/model/model.py:55-66,196-211,255,361)."""
    _need_gpu()
    from vlsat_amd import VLSATConfig, synth
    from vlsat_amd.checkpoint import save_reference_checkpoint
    from vlsat_amd.model import VLSATModel
    cfg = VLSATConfig(N_LAYERS=2)
    d = str(tmp_path / "ckp" / "Mmgnet" / "exp")
    save_reference_checkpoint(d, synth.make_weights(cfg), best=True, iteration=1234, eva_res=56.5)
    m = VLSATModel(cfg, DEV)
    assert m.load(d, best=True) and m.eval() is m and m.to(DEV) is m
    assert (m.iteration, m.eva_res) == (1234, 56.5)
    b = synth.make_batch(1, 8, 256, seed0=1000)
    x = {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}
    g = torch.Generator().manual_seed(5)
    gt_cls = torch.randint(0, 160, (8,), generator=g).to(DEV)
    gt_rel = (torch.rand(56, 26, generator=g) < 0.1).long().to(DEV)
    out = m.process_val(x["obj_points"], x["obj_2d_feats"], gt_cls, x["descriptor"], gt_rel,
                        x["edge_indices"].t().contiguous(), x["batch_ids"], use_triplet=True)
    assert len(out) == 10 and out[0].shape == (8,) and out[4].shape == out[2].shape and out[6].shape[1] == 5
    z = np.load(os.path.join(golden_dir, "cfg1_n8_p256_l2.npz"))
    o = m(x["obj_points"], x["obj_2d_feats"], x["edge_indices"], x["descriptor"], x["batch_ids"])
    assert float(np.abs(o[0].cpu().numpy() - z["obj3d"]).max()) < 1e-4
    m.close()


def _label_batches(n_batches, scenes_per_batch, seed, n_layers=1):
    """(cfg, weights, loader-style batch dicts on the GPU) with random labels; some edges carry no gt relation, some several"""
    from vlsat_amd import VLSATConfig, synth
    cfg = VLSATConfig(N_LAYERS=n_layers)
    w = synth.make_weights(cfg)
    g = torch.Generator().manual_seed(seed)
    batches = []
    for s in range(n_batches):
        sizes = [int(torch.randint(3, 12, (1,), generator=g)) for _ in range(scenes_per_batch)]
        b = synth.collate([synth.make_scene(n, 32, 9100 + 10 * s + i) for i, n in enumerate(sizes)])
        n, e = b["obj_points"].shape[0], b["edge_indices"].shape[1]
        item = {k: torch.from_numpy(v).to(DEV) for k, v in b.items() if k != "edge_indices"}
        item.update(gt_class=torch.randint(0, 160, (n,), generator=g).to(DEV),
                    gt_rel_cls=(torch.rand(e, 26, generator=g) < 0.06).long().to(DEV),
                    edge_indices=torch.from_numpy(b["edge_indices"]).t().contiguous().to(DEV), fc_sizes=sizes)
        batches.append(item)
    return cfg, w, batches


def test_device_counts_equal_the_host_accumulation():
    """vlsat_eval_counts (rank tables -> the 361 additive counts, on the device) against evaluate.accumulate (numpy, pinned to
    the reference's functions in tests/test_evaluate_cpu.py) on the SAME forward outputs: integer counts, exactly equal."""
    _need_gpu()
    from vlsat_amd import evaluate as EV, metrics as M
    from vlsat_amd.model import VLSATModel
    cfg, w, batches = _label_batches(3, 2, 11)
    model = VLSATModel(cfg, DEV).load_state(w).eval()
    vec = np.zeros(len(EV.fields()))
    counts = torch.zeros(len(EV.fields()), dtype=torch.int64, device=DEV)
    for b in batches:
        out = M.process_val(model, b["obj_points"], b["obj_2d_feats"], b["gt_class"], b["descriptor"], b["gt_rel_cls"],
                            b["edge_indices"], b["batch_ids"], use_triplet=True)
        EV.accumulate(vec, dict(top_k_obj=out[0], top_k_obj_2d=out[1], top_k_rel=out[2], top_k_rel_2d=out[3],
                                top_k_triplet=out[4], top_k_triplet_2d=out[5]), out[6], 2)
        M.process_val_counts(model, counts, b["obj_points"], b["obj_2d_feats"], b["gt_class"], b["descriptor"], b["gt_rel_cls"],
                             b["edge_indices"], b["batch_ids"], 2)
    got = counts.cpu().numpy().astype(np.float64)
    bad = [(f, g, v) for f, g, v in zip(EV.fields(), got, vec) if g != v]
    assert not bad, bad[:10]
    assert got[0] == 6 and got.sum() > 1000


def test_device_counts_no_relations_and_single_label():
    """edges without any gt relation (one 'no relation' slot each, predicate -1) and the single-label setting"""
    _need_gpu()
    from vlsat_amd import VLSATConfig, synth, evaluate as EV, metrics as M
    from vlsat_amd.model import VLSATModel
    cfg = VLSATConfig(N_LAYERS=1, multi_rel_outputs=False, num_rel_class=27)
    w = synth.make_weights(cfg)
    model = VLSATModel(cfg, DEV).load_state(w).eval()
    b = synth.collate([synth.make_scene(6, 32, 77)])
    g = torch.Generator().manual_seed(5)
    n, e = 6, 30
    item = {k: torch.from_numpy(v).to(DEV) for k, v in b.items() if k != "edge_indices"}
    edges = torch.from_numpy(b["edge_indices"]).t().contiguous().to(DEV)
    gt_cls = torch.randint(0, 160, (n,), generator=g).to(DEV)
    gt_rel = torch.randint(0, 27, (e,), generator=g)
    gt_rel[::3] = 0                                                   # 'none'
    gt_rel = gt_rel.to(DEV)
    out = M.process_val(model, item["obj_points"], item["obj_2d_feats"], gt_cls, item["descriptor"], gt_rel, edges, item["batch_ids"], use_triplet=True)
    vec = EV.accumulate(np.zeros(len(EV.fields(27))), dict(top_k_obj=out[0], top_k_obj_2d=out[1], top_k_rel=out[2], top_k_rel_2d=out[3],
                                                           top_k_triplet=out[4], top_k_triplet_2d=out[5]), out[6], 1, n_rel=27)
    counts = torch.zeros(len(EV.fields(27)), dtype=torch.int64, device=DEV)
    M.process_val_counts(model, counts, item["obj_points"], item["obj_2d_feats"], gt_cls, item["descriptor"], gt_rel, edges, item["batch_ids"], 1)
    assert np.array_equal(counts.cpu().numpy().astype(np.float64), vec)


@pytest.mark.parametrize("workers", [1, 3])
def test_pipelined_validation_equals_the_reference_compatible_loop(workers):
    """evaluate.validation(workers=K): one scene per call (validation()'s batch_size = 1), K scenes in flight on K streams and
    model replicas, counts on the device -- the same summary, to the last count, as the host-synchronous loop."""
    _need_gpu()
    from vlsat_amd import evaluate as EV
    from vlsat_amd.model import VLSATModel
    cfg, w, batches = _label_batches(14, 1, 23)
    model = VLSATModel(cfg, DEV).load_state(w).eval()
    want = EV.validation(model, batches, device=DEV)
    got = EV.validation(model, batches, device=DEV, workers=workers)
    assert got["scenes"] == 14 and got == want
    for b in batches:                                                 # without the fully-connected hint: the edge list is hashed
        b.pop("fc_sizes")
    assert EV.validation(model, batches, device=DEV, workers=workers) == want


def test_merged_validation_matches_the_one_scene_loop():
    """evaluate.validation(workers=2, merge=4): four consecutive one-scene items collated on the device per call.  Counts are
    additive; the batched forward differs from the one-scene forward in the last bits, so a near-tie may move a rank by one:
    the totals (scenes, objects, predicates, triplets, per-class sample counts) are exact, the hit counts within a few units."""
    _need_gpu()
    from vlsat_amd import evaluate as EV
    from vlsat_amd.model import VLSATModel
    cfg, w, batches = _label_batches(13, 1, 29)
    model = VLSATModel(cfg, DEV).load_state(w).eval()
    want = EV._validation_pipelined(model, batches, DEV, 1)
    F = EV.fields()
    totals = [i for i, f in enumerate(F) if f == "scenes" or "_n_" in f]

    def check(got):
        assert np.array_equal(got[totals], want[totals]) and got[F.index("scenes")] == 13
        assert np.abs(got - want).sum() <= 8, [(F[i], want[i], got[i]) for i in np.nonzero(got != want)[0]]
    check(EV._validation_pipelined(model, batches, DEV, 2, merge=4))
    s1 = EV.validation(model, batches, device=DEV, workers=2, merge=4)
    assert s1["scenes"] == 13
    for b in batches:
        b.pop("fc_sizes")
    check(EV._validation_pipelined(model, batches, DEV, 1, merge=5))


@pytest.mark.parametrize("regime", ["flat", "peaked", "ties", "few_classes", "one_hot"])
def test_triplet_staircase_counts_equal_the_brute_force_oracle(regime):
    """The triplet ranks come from a staircase walk over each node's sorted class probabilities (csrc/eval_ranks.hip) instead of
    the C x C x R outer product the reference sorts (eva_utils_acc.py:161-178).  Bit-exact against the oracle's brute-force
    counts in every shape the staircase can take: flat scores (count capped in row 0), peaked ones (a few cells), exact ties
    between classes and between triples, fewer classes than topk (K = C), probabilities that are exactly 0 and 1; edges with no,
    one, several and all relations labelled; self loops; R = 27."""
    _need_gpu()
    from vlsat_amd import metrics as M
    from oracle import metrics_oracle as MO
    g = torch.Generator().manual_seed({"flat": 1, "peaked": 2, "ties": 3, "few_classes": 4, "one_hot": 5}[regime])
    n, e, C, R = 40, 260, 160, 27
    if regime == "few_classes":
        C = 9
    scale = {"flat": 0.05, "peaked": 9.0, "ties": 3.0, "few_classes": 2.0, "one_hot": 60.0}[regime]
    logits = torch.randn(n, C, generator=g) * scale
    if regime == "ties":
        logits = torch.round(logits)                                     # many classes share a score exactly
        logits[::4] = logits[0]                                          # and whole nodes share a row
    gt = torch.randint(0, C, (n,), generator=g)
    edges = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)], 1)
    edges[7] = torch.tensor([3, 3])
    rel = torch.sigmoid(torch.randn(e, R, generator=g) * (0.2 if regime == "flat" else 4.0))
    if regime == "ties":
        rel = torch.round(rel * 8) / 8
    gt_rel = (torch.rand(e, R, generator=g) < 0.06).long()
    gt_rel[5] = 1
    gt_rel[6] = 0
    probs = F.softmax(logits, dim=-1)
    r = M.eval_ranks(logits.to(DEV), rel.to(DEV), gt.to(DEV), gt_rel.to(DEV), edges.to(DEV), obj_probs=probs.to(DEV))
    torch.cuda.synchronize()
    obj = MO.topk_object(logits, gt, 11)
    tri, _ = MO.triplet_topk(logits, rel, gt, gt_rel, edges, 101, obj, obj_probs=probs)
    got = r["top_k_triplet"].cpu().numpy()
    assert got.shape == tri.shape and np.array_equal(got, tri), (regime, int((got != tri).sum()))
    assert np.array_equal(r["top_k_rel"].cpu().numpy(), MO.topk_predicate(rel, gt_rel, 6))
    if regime in ("flat",):
        assert (got >= 90).mean() > 0.5                                  # the cap is what this regime exercises
    if regime in ("peaked", "one_hot"):
        assert (got <= 50).mean() > 0.05                                 # and real counts what this one does
