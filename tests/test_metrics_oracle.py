"""The metrics oracle (oracle/metrics_oracle.py) against goldens produced by the reference's own
ranking functions (tests/golden/make_golden_metrics.py).  CPU only, bit-exact (integer ranks)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_ranks_match_reference(golden_dir, case):
    z = np.load(os.path.join(golden_dir, "metrics_small.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    for suffix in ("", "_2d"):
        obj = MO.topk_object(g("obj_logits" + suffix), g("gt_cls"), 11)
        assert np.array_equal(obj, z[f"{case}.top_k_obj{suffix}"])
        rel = MO.topk_predicate(g("rel" + suffix), g("gt_rel"), 6)
        assert np.array_equal(rel, z[f"{case}.top_k_rel{suffix}"])
    obj3 = MO.topk_object(g("obj_logits"), g("gt_cls"), 11)
    tri, cm = MO.triplet_topk(g("obj_logits"), g("rel"), g("gt_cls"), g("gt_rel"), g("edges"), 101, obj3)
    assert np.array_equal(tri, z[f"{case}.top_k_triplet"])
    assert np.array_equal(cm, z[f"{case}.cls_matrix"])
    tri2, _ = MO.triplet_topk(g("obj_logits_2d"), g("rel_2d"), g("gt_cls"), g("gt_rel"), g("edges"), 101, obj3)
    assert np.array_equal(tri2, z[f"{case}.top_k_triplet_2d"])
    assert np.allclose(MO.mean_recall(tri, cm), z[f"{case}.mean_recall"])
    assert int(z[f"{case}.n_scores"][0]) == int((cm[:, -1] != -1).sum())


@pytest.mark.parametrize("case", ["a", "b"])
def test_single_label_ranks_match_reference(golden_dir, case):
    """multi_rel_outputs=False: label targets (0 = none), log_softmax predictions, exp() before the triple scores."""
    z = np.load(os.path.join(golden_dir, "metrics_single_label.npz"))
    g = lambda k: torch.from_numpy(z[f"{case}.{k}"])
    obj = MO.topk_object(g("obj_logits"), g("gt_cls"), 11)
    assert np.array_equal(obj, z[f"{case}.top_k_obj"])
    rel, tri, cm = MO.single_label_ranks(g("obj_logits"), g("rel"), g("gt_cls"), g("gt_rel"), g("edges"), 6, 101, obj)
    assert np.array_equal(rel, z[f"{case}.top_k_rel"])
    assert np.array_equal(tri, z[f"{case}.top_k_triplet"])
    assert np.array_equal(cm, z[f"{case}.cls_matrix"])
