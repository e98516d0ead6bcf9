"""Scene-sharded evaluation loop: the MI355X counterpart of ``MMGNet.validation`` (reference
``src/model/model.py:181-362``) for the part that sits on top of the hot path.

Each rank walks its contiguous shard of the scene list, runs forward + ranking on its GPU
(``metrics.process_val``), and accumulates a fixed-length vector of hit COUNTS.  Counts are
additive over scenes, so ONE all-reduce(SUM) at the end (RCCL over xGMI; gloo in the CPU test)
gives exactly the numbers a single process would get from the concatenated rank lists
(``validation()`` concatenates and thresholds them, :214-242, :267-282); the per-predicate
counts reproduce ``get_mean_recall`` (eva_utils_acc.py:224-237) and ``compute_mean_predicate``
(model.py:364-388)."""
from __future__ import annotations

from typing import Dict, Iterable

import numpy as np
import torch

from . import dist as vdist

N_REL = 26
# layout of the metrics vector (fp64): see fields()
_OBJ_K, _REL_K, _TRI_K = (1, 5, 10), (1, 3, 5), (50, 100)


def fields(n_rel: int = N_REL):
    f = ["scenes"]
    # get_mean_recall loops `range(int(cls_matrix.max()))` with the max taken over the WHOLE matrix
    # (object classes and ranks included): class c counts iff some entry >= c+1.  Kept additive:
    f += [f"cm_ge{k}" for k in range(1, n_rel + 1)]
    for br in ("3d", "2d"):
        f += [f"obj_n_{br}"] + [f"obj_hit@{k}_{br}" for k in _OBJ_K]
        f += [f"rel_n_{br}"] + [f"rel_hit@{k}_{br}" for k in _REL_K]
        f += [f"tri_n_{br}"] + [f"tri_hit@{k}_{br}" for k in _TRI_K]
        for c in range(n_rel):
            f += [f"cls{c}_n_{br}"] + [f"cls{c}_tri@{k}_{br}" for k in _TRI_K] + [f"cls{c}_rel@{k}_{br}" for k in _REL_K]
    return f


_IDX_CACHE: Dict[int, Dict[str, int]] = {}


def _index(n_rel: int) -> Dict[str, int]:
    if n_rel not in _IDX_CACHE:
        _IDX_CACHE[n_rel] = {k: i for i, k in enumerate(fields(n_rel))}
    return _IDX_CACHE[n_rel]


def accumulate(vec: np.ndarray, ranks: Dict[str, np.ndarray], cls_matrix: np.ndarray, n_scenes: int, n_rel: int = N_REL):
    """Add one batch's rank arrays (process_val outputs) into the counts vector (in place).  Histogram form
    (bincount per predicate class) of the per-class loops of get_mean_recall / compute_mean_predicate."""
    idx = _index(n_rel)
    vec[idx["scenes"]] += n_scenes
    cm = np.asarray(cls_matrix, dtype=np.int64).reshape(-1, 5) if len(cls_matrix) else np.zeros((0, 5), np.int64)
    pred = cm[:, -1]
    if len(cm):                                      # #entries >= k for k = 1..n_rel, from one histogram of the matrix
        h = np.bincount(np.clip(cm.ravel(), 0, n_rel), minlength=n_rel + 1)
        ge = np.cumsum(h[::-1])[::-1]                # ge[k] = #entries >= k (entries above n_rel were clipped to n_rel)
        for k in range(1, n_rel + 1):
            vec[idx[f"cm_ge{k}"]] += int(ge[k])
    has = pred >= 0
    pc = pred[has]
    for br, o, r, t in (("3d", ranks["top_k_obj"], ranks["top_k_rel"], ranks["top_k_triplet"]),
                        ("2d", ranks["top_k_obj_2d"], ranks["top_k_rel_2d"], ranks["top_k_triplet_2d"])):
        o, r, t = np.asarray(o), np.asarray(r), np.asarray(t)
        vec[idx[f"obj_n_{br}"]] += len(o)
        vec[idx[f"rel_n_{br}"]] += len(r)
        vec[idx[f"tri_n_{br}"]] += len(t)
        for k in _OBJ_K:
            vec[idx[f"obj_hit@{k}_{br}"]] += int((o <= k).sum())
        for k in _REL_K:
            vec[idx[f"rel_hit@{k}_{br}"]] += int((r <= k).sum())
        for k in _TRI_K:
            vec[idx[f"tri_hit@{k}_{br}"]] += int((t <= k).sum())
        # rows of cls_matrix align with the rank lists entry by entry; predicate -1 = "no relation" rows
        n_c = np.bincount(pc, minlength=n_rel)
        tri_c = {k: np.bincount(pc, weights=(t[has] <= k), minlength=n_rel) for k in _TRI_K}
        rel_c = {k: np.bincount(pc, weights=(r[has] <= k), minlength=n_rel) for k in _REL_K}
        for c in np.nonzero(n_c)[0]:
            if c >= n_rel:
                continue
            vec[idx[f"cls{c}_n_{br}"]] += int(n_c[c])
            for k in _TRI_K:
                vec[idx[f"cls{c}_tri@{k}_{br}"]] += int(tri_c[k][c])
            for k in _REL_K:
                vec[idx[f"cls{c}_rel@{k}_{br}"]] += int(rel_c[k][c])
    return vec


def summarize(vec: np.ndarray, n_rel: int = N_REL) -> Dict[str, float]:
    """Percentages validation() prints, from the (all-reduced) counts."""
    idx = _index(n_rel)
    out = {"scenes": float(vec[idx["scenes"]])}
    for br in ("3d", "2d"):
        for name, ks in (("obj", _OBJ_K), ("rel", _REL_K), ("tri", _TRI_K)):
            n = max(vec[idx[f"{name}_n_{br}"]], 1)
            for k in ks:
                out[f"{name}_acc@{k}_{br}"] = float(vec[idx[f"{name}_hit@{k}_{br}"]] * 100 / n)
        present = [c for c in range(n_rel) if vec[idx[f"cls{c}_n_{br}"]] > 0]
        for k in _TRI_K:                             # classes c < cls_matrix.max() (reference quirk, kept)
            rec = [vec[idx[f"cls{c}_tri@{k}_{br}"]] * 100 / vec[idx[f"cls{c}_n_{br}"]] for c in present
                   if vec[idx[f"cm_ge{c + 1}"]] > 0]
            out[f"mean_recall@{k}_{br}"] = float(np.mean(np.array(rec, dtype=np.float32))) if rec else 0.0
        for k in _REL_K:                             # compute_mean_predicate: all 26 classes with samples
            acc = [vec[idx[f"cls{c}_rel@{k}_{br}"]] / vec[idx[f"cls{c}_n_{br}"]] for c in present]
            out[f"mean_rel_acc@{k}_{br}"] = float(np.mean(acc) * 100) if acc else 0.0
    return out


_warned_sync = False


def _n_scenes(b) -> int:
    """Scenes in a batch.  ``n_scenes`` / ``fc_sizes`` are what the pipelined loop (validation(workers >= 1)) needs to stay free
    of host round trips: without either the count is read back from ``batch_ids`` on the device -- a blocking copy per batch on
    the worker's stream (and the forward then also reads the edge list back to key its plan).  Correct, but the loop runs at the
    latency of every scene; said once."""
    if "n_scenes" in b:
        return int(b["n_scenes"])
    if "fc_sizes" in b:
        return len(b["fc_sizes"])
    if not b["batch_ids"].numel():
        return 0
    if b["batch_ids"].is_cuda:
        global _warned_sync
        if not _warned_sync:
            _warned_sync = True
            import warnings
            warnings.warn("evaluate: batch without 'n_scenes' / 'fc_sizes' -- the scene count (and the plan key) are read back from "
                          "the device for every batch, which serialises the loop; add one of the hints to the loader's items",
                          RuntimeWarning, stacklevel=3)
    return int(b["batch_ids"].max().item()) + 1


def merge_batches(bs) -> dict:
    """Consecutive batches (the reference loader's item names) collated into ONE on the device: node offsets applied to
    ``edge_indices``, scene offsets to ``batch_ids``, ``fc_sizes`` hints concatenated when every batch carries one (what
    collate_fn_mmg does for a bigger batch_size, reference DataLoader.py:160-172).  Counts are additive over scenes, so a loop
    may hand the library several of its one-scene batches at a time."""
    if len(bs) == 1:
        return bs[0]
    ei, bid, n_off, s_off = [], [], 0, 0
    for b in bs:
        n = b["obj_points"].shape[0]
        ids = b.get("batch_ids")
        ids = torch.zeros(n, 1, dtype=torch.int64, device=b["obj_points"].device) if ids is None else ids.view(-1, 1)
        ei.append(b["edge_indices"] + n_off)
        bid.append(ids + s_off)
        n_off += n
        s_off += _n_scenes(b)
    out = {k: torch.cat([b[k] for b in bs]) for k in ("obj_points", "obj_2d_feats", "descriptor", "gt_class", "gt_rel_cls")}
    out["edge_indices"], out["batch_ids"], out["n_scenes"] = torch.cat(ei), torch.cat(bid), s_off
    if all("fc_sizes" in b for b in bs):
        out["fc_sizes"] = [int(x) for b in bs for x in b["fc_sizes"]]
    return out


@torch.no_grad()
def _validation_pipelined(model, batches: Iterable[dict], device, workers: int, merge: int = 1) -> np.ndarray:
    """The counts vector of this rank's batches with NO host round trip per batch and ``workers`` batches in flight:
    every worker thread owns a replica of the model (its own library handle: a handle is driven by one host thread and one
    stream at a time) and a stream; forward, ranking and counting of a batch are enqueued back to back
    (``metrics.process_val_counts``) and the counts accumulate on the device with integer atomics.  One scene per batch --
    validation()'s own pattern (batch_size = 1, reference src/model/model.py:185,201-211) -- leaves most of the 256 CUs idle
    when scenes run one after the other (a 40-object forward is ~115 dependent kernels of 5-20 us); several in flight fill
    them.  The host side scales too: the library call that enqueues a forward releases the GIL (ctypes).
    ``merge`` > 1: every worker takes that many consecutive batches at a time and collates them on the device into one call
    (``merge_batches``): the forward then runs at its batched rate.  Counts are additive, so the summary is the same up to
    near-ties (a batched forward differs from a one-scene forward in the last bits: a rank may move by one)."""
    import threading
    from . import metrics as M
    dev = torch.device(device)
    models = [model] + model.replicas(workers - 1)        # (kept by the model: building one uploads and prepares every weight)
    counts = [torch.zeros(len(fields()), dtype=torch.int64, device=dev) for _ in range(workers)]
    it, lock, errors = iter(batches), threading.Lock(), []

    def work(k):
        try:
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(device=dev) if workers > 1 else torch.cuda.current_stream(dev)
            with torch.cuda.stream(stream):
                while not errors:
                    with lock:
                        group = [b for b in (next(it, None) for _ in range(max(1, merge))) if b is not None]
                    if not group:
                        break
                    b = merge_batches(group)
                    M.process_val_counts(models[k], counts[k], b["obj_points"], b["obj_2d_feats"], b["gt_class"], b["descriptor"],
                                         b["gt_rel_cls"], b["edge_indices"], b.get("batch_ids"), _n_scenes(b), b.get("fc_sizes"))
            stream.synchronize()
        except BaseException as ex:             # (re-raised in the caller's thread)
            errors.append(ex)

    if workers == 1:
        work(0)
    else:
        cur = torch.cuda.current_stream(dev)
        cur.synchronize()                       # inputs produced on the caller's stream are complete before the workers read them
        ts = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(workers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    if errors:
        raise errors[0]
    return torch.stack(counts).sum(0).cpu().numpy().astype(np.float64)


@torch.no_grad()
def validation(model, batches: Iterable[dict], device=None, workers: int = 0, merge: int = 1) -> Dict[str, float]:
    """``batches`` yields this rank's dicts with the reference loader's item names
    (obj_points [N,3,P], obj_2d_feats, gt_class, gt_rel_cls, edge_indices [E,2], descriptor, batch_ids; optionally
    ``fc_sizes``: objects per scene when edge_indices is the canonical fully-connected list, which spares the graph plan a
    read-back of the edge list).  One collective at the very end.
    workers = 0: the reference-compatible path -- ``process_val`` per batch (numpy rank lists on the host, like
    ``Mmgnet.process_val``) and host-side accumulation.  workers >= 1: counts accumulated on the device, no host round trip
    per batch, ``workers`` batches in flight on as many streams and model replicas (for one-scene-per-call loops);
    ``merge`` = B > 1 additionally collates B consecutive batches into one call (``merge_batches``)."""
    from . import metrics as M
    if workers > 0:
        if device is None:
            raise ValueError("validation(workers > 0) needs the device")
        vec = _validation_pipelined(model, batches, device, int(workers), int(merge))
        t = vdist.allreduce_metrics(torch.from_numpy(vec).to(device))
        return summarize(t.cpu().numpy())
    vec = np.zeros(len(fields()), dtype=np.float64)
    for b in batches:
        out = M.process_val(model, b["obj_points"], b["obj_2d_feats"], b["gt_class"], b["descriptor"], b["gt_rel_cls"],
                            b["edge_indices"], b["batch_ids"], use_triplet=True)
        ranks = dict(top_k_obj=out[0], top_k_obj_2d=out[1], top_k_rel=out[2], top_k_rel_2d=out[3],
                     top_k_triplet=out[4], top_k_triplet_2d=out[5])
        n_scenes = int(b["batch_ids"].max().item()) + 1 if b["batch_ids"].numel() else 0
        accumulate(vec, ranks, out[6], n_scenes)
    t = torch.from_numpy(vec)
    if device is not None:
        t = t.to(device)
    t = vdist.allreduce_metrics(t)
    return summarize(t.cpu().numpy())
