"""Eval ranking step on the GPU: host-side mirror of what ``Mmgnet.process_val`` does with the
forward's outputs (reference ``src/model/SGFN_MMG/model.py:458-480``) and of the accuracy
summaries ``MMGNet.validation`` derives from the rank lists (reference ``src/model/model.py:227-242,
364-388``).  Ranks come from counting kernels in libvlsat_hip.so (``csrc/eval_ranks.hip``); the
reference does four ``.cpu()`` syncs and per-edge 665 600-element sorts here."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np
import torch

from . import lib as L

TOPK_OBJ, TOPK_REL, TOPK_TRIPLET, THRESHOLD = 11, 6, 101, 0.5     # process_val's constants (:463-472)


def softmax_rows(x: torch.Tensor) -> torch.Tensor:
    lib = L.load()
    x = x.contiguous()
    out = torch.empty_like(x)
    L.check(lib.vlsat_k_softmax_rows(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), L.stream_ptr()))
    return out


def multihot_targets(gt_rel: torch.Tensor, n_rel: int) -> torch.Tensor:
    """The relation target as the multi-hot [E,R] matrix the ranking kernels take.  multi_rel_outputs=True: it
    already is one.  Single-label setting (gt_rel [E] int, 0 = 'none'): get_gt keeps only labels > 0
    (reference eva_utils_acc.py:19-22), i.e. a one-hot row with column 0 cleared."""
    if gt_rel.dim() == 2:
        return gt_rel
    hot = torch.nn.functional.one_hot(gt_rel.long().view(-1), n_rel)
    hot[:, 0] = 0
    return hot


def eval_ranks(obj_logits: torch.Tensor, rel_probs: torch.Tensor, gt_class: torch.Tensor, gt_rel: torch.Tensor,
               edges: torch.Tensor, obj_probs: torch.Tensor | None = None,
               multi_rel_outputs: bool = True, rel_exp: torch.Tensor | None = None) -> Dict[str, torch.Tensor]:
    """Device tensors in, device tensors out (no sync).  ``edges`` is [E,2] (from, to) like the
    ``edge_indices`` the reference hands to process_val; ``gt_rel`` the multi-hot [E,R] target, or the [E] label
    vector of the single-label setting (multi_rel_outputs=False: ``rel_probs`` are then log-probabilities, ranked
    as they are by evaluate_topk_predicate and exponentiated by evaluate_triplet_topk, eva_utils_acc.py:146-147).
    Returns flat rank tensors in the reference's order plus ``cnt`` (ranks per edge)."""
    lib = L.load()
    n, c = obj_logits.shape
    e, r = rel_probs.shape
    dev = obj_logits.device
    obj_logits, rel_probs = obj_logits.contiguous(), rel_probs.contiguous()
    gt_class = gt_class.to(torch.int64).contiguous().view(-1)
    gt_rel = multihot_targets(gt_rel, r).to(torch.int64).contiguous()
    edges = edges.to(torch.int64).contiguous()
    if edges.shape != (e, 2) or gt_rel.shape != (e, r) or gt_class.numel() != n:
        raise L.VlsatError("eval_ranks: edges must be [E,2], gt_rel [E,R], gt_class [N]")
    if obj_probs is None:
        obj_probs = softmax_rows(obj_logits)
    obj_rank = torch.empty(n, dtype=torch.int32, device=dev)
    rel_rank = torch.empty(e, r, dtype=torch.int32, device=dev)
    tri_rank = torch.empty(e, r, dtype=torch.int32, device=dev)
    cnt = torch.empty(e, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(int(lib.vlsat_eval_ranks_scratch_floats(n, c, TOPK_TRIPLET)), 1), dtype=torch.float32, device=dev)
    L.check(lib.vlsat_eval_ranks(obj_logits.data_ptr(), obj_probs.data_ptr(), rel_probs.data_ptr(), gt_class.data_ptr(),
                                 gt_rel.data_ptr(), edges.data_ptr(), n, e, c, r, TOPK_OBJ, TOPK_REL, TOPK_TRIPLET,
                                 THRESHOLD, obj_rank.data_ptr(), rel_rank.data_ptr(), tri_rank.data_ptr(), cnt.data_ptr(),
                                 scratch.data_ptr(), L.stream_ptr()))
    if not multi_rel_outputs:            # triplet scores use exp(log_softmax); predicate ranks above used the raw values
        tri_rank = torch.empty(e, r, dtype=torch.int32, device=dev)
        rel_exp = (rel_probs.exp() if rel_exp is None else rel_exp).contiguous()
        L.check(lib.vlsat_eval_ranks(obj_logits.data_ptr(), obj_probs.data_ptr(), rel_exp.data_ptr(), gt_class.data_ptr(),
                                     gt_rel.data_ptr(), edges.data_ptr(), n, e, c, r, TOPK_OBJ, TOPK_REL, TOPK_TRIPLET,
                                     THRESHOLD, obj_rank.data_ptr(), torch.empty_like(rel_rank).data_ptr(),
                                     tri_rank.data_ptr(), torch.empty_like(cnt).data_ptr(), scratch.data_ptr(), L.stream_ptr()))
    used = torch.arange(r, device=dev)[None, :] < cnt[:, None]          # first cnt[e] slots of each edge row
    return {"top_k_obj": obj_rank, "top_k_rel": rel_rank[used], "top_k_triplet": tri_rank[used], "cnt": cnt,
            "obj_probs": obj_probs}


def rank_tables(obj_logits: torch.Tensor, rel_probs: torch.Tensor, gt_class: torch.Tensor, gt_rel: torch.Tensor,
                edges: torch.Tensor, multi_rel_outputs: bool = True) -> Dict[str, torch.Tensor]:
    """``eval_ranks`` without its variable-length outputs: the rank TABLES as the kernels write them -- obj_rank [N],
    rel_rank / tri_rank [E,R] (first cnt[e] slots of row e used), cnt [E] -- all on the device, nothing that depends on
    their values (no boolean indexing, hence no host synchronisation).  Inputs must already be contiguous, gt_rel the
    multi-hot int64 [E,R] target, edges int64 [E,2].  What ``eval_counts`` consumes."""
    lib = L.load()
    n, c = obj_logits.shape
    e, r = rel_probs.shape
    dev = obj_logits.device
    obj_probs = softmax_rows(obj_logits)
    obj_rank = torch.empty(n, dtype=torch.int32, device=dev)
    rel_rank = torch.empty(e, r, dtype=torch.int32, device=dev)
    tri_rank = torch.empty(e, r, dtype=torch.int32, device=dev)
    cnt = torch.empty(e, dtype=torch.int32, device=dev)
    scratch = torch.empty(max(int(lib.vlsat_eval_ranks_scratch_floats(n, c, TOPK_TRIPLET)), 1), dtype=torch.float32, device=dev)
    L.check(lib.vlsat_eval_ranks(obj_logits.data_ptr(), obj_probs.data_ptr(), rel_probs.data_ptr(), gt_class.data_ptr(),
                                 gt_rel.data_ptr(), edges.data_ptr(), n, e, c, r, TOPK_OBJ, TOPK_REL, TOPK_TRIPLET,
                                 THRESHOLD, obj_rank.data_ptr(), rel_rank.data_ptr(), tri_rank.data_ptr(), cnt.data_ptr(),
                                 scratch.data_ptr(), L.stream_ptr()))
    if not multi_rel_outputs:            # triplet scores use exp(log_softmax); the predicate ranks above used the raw values
        tri_rank = torch.empty(e, r, dtype=torch.int32, device=dev)
        rel_exp = rel_probs.exp()
        L.check(lib.vlsat_eval_ranks(obj_logits.data_ptr(), obj_probs.data_ptr(), rel_exp.data_ptr(), gt_class.data_ptr(),
                                     gt_rel.data_ptr(), edges.data_ptr(), n, e, c, r, TOPK_OBJ, TOPK_REL, TOPK_TRIPLET,
                                     THRESHOLD, torch.empty_like(obj_rank).data_ptr(), torch.empty_like(rel_rank).data_ptr(),
                                     tri_rank.data_ptr(), torch.empty_like(cnt).data_ptr(), scratch.data_ptr(), L.stream_ptr()))
    return {"obj_rank": obj_rank, "rel_rank": rel_rank, "tri_rank": tri_rank, "cnt": cnt}


def eval_counts(counts: torch.Tensor, t3: Dict[str, torch.Tensor], t2: Dict[str, torch.Tensor], gt_class: torch.Tensor,
                gt_rel: torch.Tensor, edges: torch.Tensor, n_scenes: int) -> torch.Tensor:
    """counts (device int64 [len(evaluate.fields())], zeroed once by the caller) += the additive counts of one batch, from the
    rank tables of its 3D (``t3``) and 2D (``t2``) outputs: ``vlsat_eval_counts`` -- the device twin of
    ``evaluate.accumulate`` (integer atomics: exact, order-independent, safe from several streams).  No synchronisation."""
    lib = L.load()
    e, r = t3["rel_rank"].shape
    n = t3["obj_rank"].numel()
    if counts.dtype != torch.int64 or counts.numel() != 1 + r + 2 * (11 + 6 * r) or not counts.is_contiguous():
        raise L.VlsatError("eval_counts: counts must be a contiguous int64 vector of 1 + R + 2 (11 + 6 R) entries")
    L.check(lib.vlsat_eval_counts(t3["obj_rank"].data_ptr(), t2["obj_rank"].data_ptr(), t3["rel_rank"].data_ptr(), t2["rel_rank"].data_ptr(),
                                  t3["tri_rank"].data_ptr(), t2["tri_rank"].data_ptr(), t3["cnt"].data_ptr(), gt_class.data_ptr(),
                                  gt_rel.data_ptr(), edges.data_ptr(), n, e, r, int(n_scenes), counts.data_ptr(), L.stream_ptr()))
    return counts


@torch.no_grad()
def process_val_counts(model, counts: torch.Tensor, obj_points, obj_2d_feats, gt_cls, descriptor, gt_rel_cls, edge_indices,
                       batch_ids=None, n_scenes: int = 1, fc_sizes=None):
    """``process_val`` for an evaluation LOOP: forward + both ranking passes + the counts, all enqueued on the current stream,
    nothing read back (``Mmgnet.process_val`` returns numpy rank lists, i.e. four host round trips per scene, reference
    SGFN_MMG/model.py:463-480; ``validation()`` only ever turns them into the counts accumulated here).
    ``edge_indices`` is [E,2] as the data loader yields it, on the device."""
    multi = bool(getattr(getattr(model, "config", None), "multi_rel_outputs", True))
    edges = edge_indices.to(torch.int64).contiguous()
    if multi and hasattr(model, "process_val_counts"):           # one library call: forward + ranking + counts in the plan's scratch
        r = model.config.num_rel_class
        if model.process_val_counts(counts, obj_points, obj_2d_feats, gt_cls.to(torch.int64).contiguous().view(-1), descriptor,
                                    multihot_targets(gt_rel_cls, r).to(torch.int64).contiguous(), edges, batch_ids, n_scenes, fc_sizes):
            return counts
    # with the fully-connected hint the plan never reads the edge list: the [2,E] view is enough (no transpose kernel)
    ei_t = edges.t() if fc_sizes is not None else edges.t().contiguous()
    obj3, obj2, rel3, rel2 = model(obj_points, obj_2d_feats, ei_t, descriptor, batch_ids, istrain=False, fc_sizes=fc_sizes)
    gt_rel = multihot_targets(gt_rel_cls, rel3.shape[1]).to(torch.int64).contiguous()
    gt_cls = gt_cls.to(torch.int64).contiguous().view(-1)
    t3 = rank_tables(obj3, rel3, gt_cls, gt_rel, edges, multi)
    t2 = rank_tables(obj2, rel2, gt_cls, gt_rel, edges, multi)
    return eval_counts(counts, t3, t2, gt_cls, gt_rel, edges, n_scenes)


def cls_matrix(gt_class: torch.Tensor, gt_rel: torch.Tensor, edges: torch.Tensor, obj_topk: torch.Tensor) -> torch.Tensor:
    """[n,5] rows (sub_gt, sub_pred_rank, obj_gt, obj_pred_rank, predicate | -1) in the order
    evaluate_triplet_topk appends them (eva_utils_acc.py:185-199): per edge its gt predicates in
    ascending class order, or one row with -1 when the edge has no gt relation."""
    if gt_rel.dim() == 1:
        raise L.VlsatError("cls_matrix: pass the multi-hot target (metrics.multihot_targets)")
    e, r = gt_rel.shape
    has = gt_rel == 1
    none = ~has.any(1)
    slot = torch.cat([none[:, None], has], 1)                             # column 0 = the "-1" row
    ei, ki = torch.nonzero(slot, as_tuple=True)                           # row-major = edge order, ascending class
    a, b = edges[ei, 0], edges[ei, 1]
    gt_class = gt_class.view(-1)
    return torch.stack([gt_class[a], obj_topk[a].long(), gt_class[b], obj_topk[b].long(), ki - 1], 1)


@torch.no_grad()
def process_val(model, obj_points, obj_2d_feats, gt_cls, descriptor, gt_rel_cls, edge_indices, batch_ids=None,
                use_triplet=True):
    """Same call and 10-tuple as ``Mmgnet.process_val`` (reference SGFN_MMG/model.py:458-480);
    ``edge_indices`` is [E,2] as the data loader yields it.  Rank arrays are numpy int64 like the
    reference's; the score lists come back stacked as tensors."""
    ei_t = edge_indices.t().contiguous()
    obj3, obj2, rel3, rel2 = model(obj_points, obj_2d_feats, ei_t, descriptor, batch_ids, istrain=False)
    multi = bool(getattr(getattr(model, "config", None), "multi_rel_outputs", True))     # self.mconfig.multi_rel_outputs
    gt_rel_cls = multihot_targets(gt_rel_cls, rel3.shape[1])
    r3 = eval_ranks(obj3, rel3, gt_cls, gt_rel_cls, edge_indices, multi_rel_outputs=multi)
    r2 = eval_ranks(obj2, rel2, gt_cls, gt_rel_cls, edge_indices, multi_rel_outputs=multi)
    cm = cls_matrix(gt_cls, gt_rel_cls, edge_indices, r3["top_k_obj"])     # obj_topk = 3D ranks for both (:469-470)
    np64 = lambda t: t.cpu().numpy().astype(np.int64)
    if not use_triplet:
        return np64(r3["top_k_obj"]), np64(r2["top_k_obj"]), np64(r3["top_k_rel"]), np64(r2["top_k_rel"]), [101], None, None, None, None, None
    has = (gt_rel_cls == 1)
    ei, _ = torch.nonzero(has, as_tuple=True)
    sub_scores = r3["obj_probs"][edge_indices[ei, 0]]
    obj_scores = r3["obj_probs"][edge_indices[ei, 1]]
    rel_scores = rel3[ei] if multi else rel3[ei].exp()
    return (np64(r3["top_k_obj"]), np64(r2["top_k_obj"]), np64(r3["top_k_rel"]), np64(r2["top_k_rel"]),
            np64(r3["top_k_triplet"]), np64(r2["top_k_triplet"]), cm.cpu().numpy(), sub_scores, obj_scores, rel_scores)


def summarize(top_k_obj, top_k_rel, top_k_triplet, cls_mat=None) -> Dict[str, float]:
    """Accuracy summaries of validation() (reference src/model/model.py:267-282) and
    get_mean_recall (eva_utils_acc.py:224-237) from the rank arrays."""
    o, r, t = np.asarray(top_k_obj), np.asarray(top_k_rel), np.asarray(top_k_triplet)
    pct = lambda a, k: float((a <= k).sum() * 100 / max(len(a), 1))
    out = {"obj_acc@1": pct(o, 1), "obj_acc@5": pct(o, 5), "obj_acc@10": pct(o, 10),
           "rel_acc@1": pct(r, 1), "rel_acc@3": pct(r, 3), "rel_acc@5": pct(r, 5),
           "triplet_acc@50": pct(t, 50), "triplet_acc@100": pct(t, 100)}
    if cls_mat is not None and len(cls_mat):
        cm = np.asarray(cls_mat)
        rec = [[], []]
        for i in range(int(cm.max())):
            sel = t[cm[:, -1] == i]
            if len(sel):
                rec[0].append((sel <= 50).sum() * 100 / len(sel))
                rec[1].append((sel <= 100).sum() * 100 / len(sel))
        if rec[0]:
            out["mean_recall@50"], out["mean_recall@100"] = float(np.mean(rec[0])), float(np.mean(rec[1]))
    return out
