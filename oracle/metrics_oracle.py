"""CPU ORACLE for the eval ranking step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, in vectorised form with identical results, the ranking functions process_val calls
right after the forward (reference src/model/SGFN_MMG/model.py:463-472):
  evaluate_topk_object     reference src/utils/eva_utils_acc.py:27-39
  get_gt                   :6-24
  evaluate_topk_predicate  :42-79
  evaluate_triplet_topk    :137-213   (use_clip=True: softmax over the object logits)
  get_mean_recall          :224-237
Parity status: PINNED by tests/test_metrics_oracle.py against tests/golden/metrics_small.npz,
produced by calling the reference functions themselves (tests/golden/make_golden_metrics.py).

Loop semantics restated as counts (proved in the docstrings): walking a descending sort until
`pred[gt] >= pred[idx] or index > topk` stops after min(#strictly-greater, topk) steps.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def topk_object(obj_pred: torch.Tensor, gt: torch.Tensor, topk: int) -> np.ndarray:
    """rank[n] = min(#{c : pred[n,c] > pred[n,gt[n]]}, topk) + 1   (eva_utils_acc.py:27-39)."""
    g = obj_pred.gather(1, gt.view(-1, 1).long())
    greater = (obj_pred > g).sum(1)
    return (torch.clamp(greater, max=topk) + 1).numpy().astype(np.int64)


def _edge_ranks(ranks_per_edge):
    """per edge: sort ascending, subtract 0,1,2,... (eva_utils_acc.py:73-77 / :206-210)."""
    out = []
    for r in ranks_per_edge:
        for c, v in enumerate(sorted(r)):
            out.append(v - c)
    return np.asarray(out, dtype=np.int64)


def topk_predicate(rel_pred: torch.Tensor, gt_rel: torch.Tensor, topk: int, thr: float = 0.5) -> np.ndarray:
    """eva_utils_acc.py:42-79 with multi_rel_outputs=True.  gt_rel is the multi-hot [E,R] target
    (get_gt lists the set bits in ascending class order, :15-17)."""
    E, R = rel_pred.shape
    per_edge = []
    for e in range(E):
        p = rel_pred[e]
        gts = torch.nonzero(gt_rel[e] == 1).view(-1).tolist()
        ranks = []
        if not gts:                                       # no gt relation (:55-61)
            ge = int((p >= thr).sum())                    # sorted desc: first index with conf < thr
            ranks.append(topk + 1 if ge == R else ge + 1)
        for k in gts:
            ranks.append(min(int((p > p[k]).sum()), topk) + 1)
        per_edge.append(ranks)
    return _edge_ranks(per_edge)


def triplet_topk(obj_logits: torch.Tensor, rel_pred: torch.Tensor, gt_cls: torch.Tensor, gt_rel: torch.Tensor,
                 edges: torch.Tensor, topk: int, obj_topk: np.ndarray, thr: float = 0.5, obj_probs=None):
    """eva_utils_acc.py:137-213 with multi_rel_outputs=True, use_clip=True.  edges [E,2] (from, to).
    conf[i,j,k] = (sub[i]*obj[j])*rel[k] (two einsums, :161-162).  Position of the gt triple in the
    descending top-`topk` list = #{conf > gt_conf} + 1 when that is <= topk, else topk+1; with no gt
    relation: min(#{conf >= thr}, topk) + 1.  Returns (res, cls_matrix [n,5])."""
    probs = F.softmax(obj_logits, dim=-1) if obj_probs is None else obj_probs
    per_edge, cls = [], []
    for e in range(edges.shape[0]):
        a, b = int(edges[e, 0]), int(edges[e, 1])
        sub, obj, rel = probs[a], probs[b], rel_pred[e]
        conf = torch.einsum("nl,m->nlm", torch.einsum("n,m->nm", sub, obj), rel).reshape(-1)
        sg, og = int(gt_cls[a]), int(gt_cls[b])
        gts = torch.nonzero(gt_rel[e] == 1).view(-1).tolist()
        ranks = []
        if not gts:
            ranks.append(min(int((conf >= thr).sum()), topk) + 1)
            cls.append([sg, int(obj_topk[a]), og, int(obj_topk[b]), -1])
        for k in gts:
            gt_conf = (sub[sg] * obj[og]) * rel[k]
            ranks.append(min(int((conf > gt_conf).sum()), topk) + 1)
            cls.append([sg, int(obj_topk[a]), og, int(obj_topk[b]), k])
        per_edge.append(ranks)
    return _edge_ranks(per_edge), np.asarray(cls, dtype=np.int64).reshape(-1, 5)


def single_label_ranks(obj_logits: torch.Tensor, rel_logp: torch.Tensor, gt_cls: torch.Tensor, gt_rel_1d: torch.Tensor,
                       edges: torch.Tensor, topk_rel: int, topk_tri: int, obj_topk: np.ndarray):
    """multi_rel_outputs=False: the target is one label per edge, 0 = 'none' (get_gt keeps labels > 0,
    eva_utils_acc.py:19-22), the head outputs log-probabilities; evaluate_topk_predicate ranks them as they are,
    evaluate_triplet_topk exponentiates them first (:146-147).  Returns (top_k_rel, top_k_triplet, cls_matrix)."""
    hot = torch.nn.functional.one_hot(gt_rel_1d.long().view(-1), rel_logp.shape[1])
    hot[:, 0] = 0
    rel_rank = topk_predicate(rel_logp, hot, topk_rel)
    tri, cm = triplet_topk(obj_logits, rel_logp.exp(), gt_cls, hot, edges, topk_tri, obj_topk)
    return rel_rank, tri, cm


def mean_recall(triplet_rank: np.ndarray, cls_matrix: np.ndarray, topk=(50, 100)) -> np.ndarray:
    """eva_utils_acc.py:224-237 (note: classes 0..max-1 only, as the reference loops range(max))."""
    if len(cls_matrix) == 0:
        return np.array([0, 0])
    rec = [[] for _ in topk]
    for i in range(int(cls_matrix.max())):
        r = triplet_rank[cls_matrix[:, -1] == i]
        if len(r) == 0:
            continue
        for j, t in enumerate(topk):
            rec[j].append((r <= t).sum() * 100 / len(r))
    return np.array(rec, dtype=np.float32).mean(axis=1)
