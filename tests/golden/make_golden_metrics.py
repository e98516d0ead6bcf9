#!/usr/bin/env python3
"""Golden vectors for the eval ranking step that follows the forward in process_val
(reference src/model/SGFN_MMG/model.py:463-472): evaluate_topk_object / evaluate_topk_predicate /
evaluate_triplet_topk / get_gt / get_mean_recall of src/utils/eva_utils_acc.py, called directly
(the module imports only numpy/torch).  Inputs are small and stored with the outputs."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from src.utils import eva_utils_acc as R  # noqa: E402


def case(seed, n, e_keep, C=160, Rn=26, sharp=6.0):
    g = torch.Generator().manual_seed(seed)
    obj_logits = torch.randn(n, C, generator=g) * sharp
    obj_logits_2d = torch.randn(n, C, generator=g) * sharp
    gt_cls = torch.randint(0, C, (n,), generator=g)
    # make some predictions right so ranks are not all at the cap
    for i in range(0, n, 2):
        obj_logits[i, gt_cls[i]] += 3 * sharp
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    pick = torch.randperm(len(pairs), generator=g)[:e_keep].tolist()
    edges = torch.tensor([pairs[i] for i in pick], dtype=torch.long)          # [E,2] like collate_fn_mmg
    E = edges.shape[0]
    rel = torch.sigmoid(torch.randn(E, Rn, generator=g) * 2.5)
    rel_2d = torch.sigmoid(torch.randn(E, Rn, generator=g) * 2.5)
    gt_rel = torch.zeros(E, Rn, dtype=torch.long)
    for e in range(E):
        k = int(torch.randint(0, 4, (1,), generator=g))                       # 0 (no relation) .. 3 labels
        if k:
            gt_rel[e, torch.randperm(Rn, generator=g)[:k]] = 1
    if E > 3:
        rel[1] = 0.9                                                          # every predicate above threshold, no gt
        gt_rel[1] = 0
        rel[2] = 0.1
        gt_rel[2] = 0
    out = {}
    top_k_obj = R.evaluate_topk_object(obj_logits, gt_cls, topk=11)
    top_k_obj_2d = R.evaluate_topk_object(obj_logits_2d, gt_cls, topk=11)
    gt_edges = R.get_gt(gt_cls, gt_rel, edges, True)
    top_k_rel = R.evaluate_topk_predicate(rel, gt_edges, True, topk=6)
    top_k_rel_2d = R.evaluate_topk_predicate(rel_2d, gt_edges, True, topk=6)
    tri, cls_matrix, ss, os_, rs = R.evaluate_triplet_topk(obj_logits, rel, gt_edges, edges, True, topk=101,
                                                          use_clip=True, obj_topk=top_k_obj)
    tri2, _, _, _, _ = R.evaluate_triplet_topk(obj_logits_2d, rel_2d, gt_edges, edges, True, topk=101,
                                               use_clip=True, obj_topk=top_k_obj)
    cm = np.array([[int(x) for x in row] for row in cls_matrix], dtype=np.int64)
    out.update(obj_logits=obj_logits.numpy(), obj_logits_2d=obj_logits_2d.numpy(), gt_cls=gt_cls.numpy(),
               edges=edges.numpy(), rel=rel.numpy(), rel_2d=rel_2d.numpy(), gt_rel=gt_rel.numpy(),
               top_k_obj=top_k_obj, top_k_obj_2d=top_k_obj_2d, top_k_rel=top_k_rel, top_k_rel_2d=top_k_rel_2d,
               top_k_triplet=tri, top_k_triplet_2d=tri2, cls_matrix=cm,
               mean_recall=R.get_mean_recall(tri, cm), n_scores=np.array([len(ss)]))
    return out


def case_single(seed, n, e_keep, C=160, Rn=27, sharp=6.0):
    """multi_rel_outputs=False: one label per edge (0 = none), log_softmax predictions."""
    g = torch.Generator().manual_seed(seed)
    obj_logits = torch.randn(n, C, generator=g) * sharp
    gt_cls = torch.randint(0, C, (n,), generator=g)
    for i in range(0, n, 2):
        obj_logits[i, gt_cls[i]] += 3 * sharp
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    pick = torch.randperm(len(pairs), generator=g)[:e_keep].tolist()
    edges = torch.tensor([pairs[i] for i in pick], dtype=torch.long)
    E = edges.shape[0]
    rel = torch.log_softmax(torch.randn(E, Rn, generator=g) * 3.0, dim=1)
    gt_rel = torch.randint(0, Rn, (E,), generator=g)
    gt_rel[torch.rand(E, generator=g) < 0.4] = 0                               # 'none'
    for e in range(0, E, 3):                                                   # some correct predictions
        if gt_rel[e] > 0:
            rel[e, gt_rel[e]] = rel[e].max() + 0.5
    top_k_obj = R.evaluate_topk_object(obj_logits, gt_cls, topk=11)
    gt_edges = R.get_gt(gt_cls, gt_rel, edges, False)
    top_k_rel = R.evaluate_topk_predicate(rel, gt_edges, False, topk=6)
    tri, cls_matrix, ss, _, _ = R.evaluate_triplet_topk(obj_logits, rel.clone(), gt_edges, edges, False, topk=101,
                                                        use_clip=True, obj_topk=top_k_obj)
    cm = np.array([[int(x) for x in row] for row in cls_matrix], dtype=np.int64)
    return dict(obj_logits=obj_logits.numpy(), gt_cls=gt_cls.numpy(), edges=edges.numpy(), rel=rel.numpy(),
                gt_rel=gt_rel.numpy(), top_k_obj=top_k_obj, top_k_rel=top_k_rel, top_k_triplet=tri, cls_matrix=cm)


def main():
    single = {}
    for name, (seed, n, e) in {"a": (11, 7, 42), "b": (12, 10, 60)}.items():
        for k, v in case_single(seed, n, e).items():
            single[f"{name}.{k}"] = v
    np.savez_compressed(os.path.join(HERE, "metrics_single_label.npz"), **single)
    allc = {}
    for name, (seed, n, e) in {"a": (1, 8, 56), "b": (2, 5, 11), "c": (3, 12, 70)}.items():
        for k, v in case(seed, n, e).items():
            allc[f"{name}.{k}"] = v
    np.savez_compressed(os.path.join(HERE, "metrics_small.npz"), **allc)
    print("written", len(allc), "arrays")


if __name__ == "__main__":
    main()
