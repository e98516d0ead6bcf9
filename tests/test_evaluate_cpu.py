"""Counts-vector aggregation of the sharded eval loop (evaluate.py): equals the reference's
list-concatenation summaries, and is invariant to how scenes are sharded over ranks (gloo, 2 ranks)."""
import os
import subprocess
import sys

import numpy as np

import vlsat_amd  # noqa: F401
from vlsat_amd import evaluate as EV

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "metrics_small.npz")


def _case(z, c):
    ranks = {k: z[f"{c}.{k}"] for k in ("top_k_obj", "top_k_obj_2d", "top_k_rel", "top_k_rel_2d", "top_k_triplet",
                                        "top_k_triplet_2d")}
    return ranks, z[f"{c}.cls_matrix"]


def test_counts_reproduce_reference_summaries():
    z = np.load(GOLD)
    vec = np.zeros(len(EV.fields()))
    for c in "abc":
        EV.accumulate(vec, *_case(z, c), n_scenes=1)
    s = EV.summarize(vec)
    cat = lambda k: np.concatenate([z[f"{c}.{k}"] for c in "abc"])
    o, r, t, cm = cat("top_k_obj"), cat("top_k_rel"), cat("top_k_triplet"), np.concatenate([z[f"{c}.cls_matrix"] for c in "abc"])
    assert s["scenes"] == 3
    assert np.isclose(s["obj_acc@1_3d"], (o <= 1).sum() * 100 / len(o))              # model.py:267-272
    assert np.isclose(s["rel_acc@3_3d"], (r <= 3).sum() * 100 / len(r))
    assert np.isclose(s["tri_acc@50_3d"], (t <= 50).sum() * 100 / len(t))
    assert np.isclose(s["tri_acc@100_2d"], (cat("top_k_triplet_2d") <= 100).sum() * 100 / len(t))
    # get_mean_recall on the concatenation (eva_utils_acc.py:224-237), restated here literally
    rec = [[], []]
    for i in range(int(cm.max())):
        sel = t[cm[:, -1] == i]
        if len(sel):
            rec[0].append((sel <= 50).sum() * 100 / len(sel))
            rec[1].append((sel <= 100).sum() * 100 / len(sel))
    mr = np.array(rec, dtype=np.float32).mean(axis=1)
    assert np.allclose([s["mean_recall@50_3d"], s["mean_recall@100_3d"]], mr)
    # single scene: equals the value the reference function itself returned for that scene
    v1 = EV.accumulate(np.zeros(len(EV.fields())), *_case(z, "a"), n_scenes=1)
    s1 = EV.summarize(v1)
    assert np.allclose([s1["mean_recall@50_3d"], s1["mean_recall@100_3d"]], z["a.mean_recall"])
    # compute_mean_predicate (model.py:364-388)
    acc = []
    for i in range(26):
        sel = r[cm[:, -1] == i]
        if len(sel):
            acc.append((sel <= 1).sum() / len(sel))
    assert np.isclose(s["mean_rel_acc@1_3d"], np.mean(acc) * 100)


_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["VLSAT_ROOT"])
import vlsat_amd
from vlsat_amd import evaluate as EV, dist as vdist
rank, local, world = vdist.init("gloo")
z = np.load(os.environ["VLSAT_GOLD"])
cases = "abc"
vec = np.zeros(len(EV.fields()))
for i in vdist.shard(len(cases), rank, world):
    c = cases[i]
    ranks = {k: z[f"{c}.{k}"] for k in ("top_k_obj", "top_k_obj_2d", "top_k_rel", "top_k_rel_2d", "top_k_triplet", "top_k_triplet_2d")}
    EV.accumulate(vec, ranks, z[f"{c}.cls_matrix"], 1)
t = vdist.allreduce_metrics(torch.from_numpy(vec))
if rank == 0:
    print("VEC", " ".join(repr(float(x)) for x in t.tolist()))
"""


def test_two_rank_allreduce_equals_single_process(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, VLSAT_ROOT=ROOT, VLSAT_GOLD=GOLD, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29641", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.array([float(x) for x in [l for l in r.stdout.splitlines() if l.startswith("VEC")][0].split()[1:]])
    z = np.load(GOLD)
    vec = np.zeros(len(EV.fields()))
    for c in "abc":
        EV.accumulate(vec, *_case(z, c), n_scenes=1)
    assert np.array_equal(got, vec)


def test_merge_batches_is_the_loaders_collation():
    """evaluate.merge_batches of one-scene items == synth.collate of the same scenes (node offsets on the edges, scene ids,
    concatenated fc_sizes hints), with and without batch_ids / hints."""
    import torch
    from vlsat_amd import synth
    scenes = [synth.make_scene(n, 16, 50 + n) for n in (3, 5, 4)]
    want = synth.collate(scenes)
    items = []
    for i, sc in enumerate(scenes):
        b = synth.collate([sc])
        n, e = b["obj_points"].shape[0], b["edge_indices"].shape[1]
        it = {k: torch.from_numpy(v) for k, v in b.items() if k != "edge_indices"}
        it.update(edge_indices=torch.from_numpy(b["edge_indices"]).t().contiguous(), gt_class=torch.full((n,), i),
                  gt_rel_cls=torch.full((e, 26), i), fc_sizes=[n])
        items.append(it)
    m = EV.merge_batches(items)
    assert m["fc_sizes"] == [3, 5, 4] and m["n_scenes"] == 3
    assert np.array_equal(m["edge_indices"].t().numpy(), want["edge_indices"])
    assert np.array_equal(m["batch_ids"].numpy().reshape(-1), want["batch_ids"].reshape(-1))
    for k in ("obj_points", "obj_2d_feats", "descriptor"):
        assert np.array_equal(m[k].numpy(), want[k])
    assert m["gt_class"].tolist() == [0] * 3 + [1] * 5 + [2] * 4 and m["gt_rel_cls"].shape == (want["edge_indices"].shape[1], 26)
    for it in items:                      # no hints, no batch_ids: scene ids are made up, the scene count comes from n_scenes
        it.pop("fc_sizes"), it.pop("batch_ids")
        it["n_scenes"] = 1
    m2 = EV.merge_batches(items)
    assert "fc_sizes" not in m2 and m2["n_scenes"] == 3 and m2["batch_ids"].view(-1).tolist() == [0] * 3 + [1] * 5 + [2] * 4
    assert EV.merge_batches(items[:1]) is items[0]
