"""Host-side mirror of the reference model object for the eval forward path.

``VLSATModel`` keeps the surface ``MMGNet.validation`` touches on ``Mmgnet`` (reference
``src/model/model.py:181-211``, ``src/model/SGFN_MMG/model.py:288-335,458-460``):
``forward`` with the same tensor-in/tensor-out signature, ``eval()``, ``to()``, and weight
loading by reference ``state_dict`` key.  All arithmetic happens in libvlsat_hip.so; this
file only validates shapes, caches the per-graph plan and passes raw pointers.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import math
import weakref
from collections import OrderedDict
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import lib as L
from .config import VLSATConfig, param_shapes

_AGGR = {"max": 0, "add": 1, "mean": 2}


class _Plan:
    def __init__(self, handle, perm, ws_bytes=0):
        self.handle = handle          # vlsat_plan (c_void_p)
        self.perm = perm              # device int64 permutation applied to the edges, or None
        self.ws_bytes = ws_bytes

    def destroy(self):
        if self.handle:
            L.load().vlsat_plan_destroy(self.handle)
            self.handle = None


class VLSATModel:
    """Drop-in for ``Mmgnet`` on the eval forward path (one instance is not re-entrant)."""

    MAX_PLANS = 64                      # cached graph plans (LRU) ...
    MAX_PLAN_BYTES = 16 << 30           # ... and the device workspace they may hold together

    def __init__(self, config: Optional[VLSATConfig] = None, device: str = "cuda:0"):
        self.config = config or VLSATConfig()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.VlsatError("VLSATModel runs on an MI355X only (device must be cuda:N); there is no CPU path")
        if not torch.cuda.is_available():
            raise L.VlsatError("no HIP device visible: libvlsat_hip.so cannot run (there is no CPU fallback)")
        self._lib = L.load()
        c = self.config
        dims = L.VlsatDims(c.N_LAYERS, c.NUM_HEADS, c.DIM_ATTEN, _AGGR[c.GCN_AGGR], c.dim_point,
                           c.num_obj_class, c.num_rel_class, float(c.obj_logit_scale), int(c.USE_GCN_EDGE),
                           int(c.multi_rel_outputs), int(c.feature_transform))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self._lib.vlsat_create(C.byref(dims), C.byref(self._h)))
        self._loaded = False
        self._zero_bid = {}
        self._plans: "OrderedDict[tuple, _Plan]" = OrderedDict()
        self._ident: "OrderedDict[tuple, tuple]" = OrderedDict()     # (id(edge tensor), id(batch tensor)) -> plan key
        self._fc_verified = set()           # fc_sizes keys whose device edge list has been checked against the canonical graph
        self.verify_fc_every_call = False   # re-check the fc_sizes claim on every call (debugging; costs a stream sync per call)
        self._debug_options = {}            # vlsat_debug_option settings applied to this handle (replayed by replicate())
        self.plan_stats = {"hits": 0, "identity_hits": 0, "builds": 0, "d2h_copies": 0}
        self.training = False
        self.gemm_precision = "fp32"
        self.batch_mode = "per_scene"
        # attributes MMGNet.validation reads on the model object (reference src/model/model.py:255,361)
        self.iteration, self.eva_res, self.epoch = 0, 0, -1

    PRECISIONS = {"fp32": 0, "bf16": 1, "bf16_mixed": 2, "bf16x3": 3, "bf16x3_attn1": 4, "fp16_mixed": 5}

    def set_gemm_precision(self, mode: str):
        """'fp32' (default, exact-fp32 MFMA: BASELINE configs[1]) | 'bf16x3' (split-bf16 MFMA, three bf16 MFMAs per
        product, fp32 accumulate, ~1e-5 error) | 'bf16_mixed' (single-rounded bf16 on the edge-row matrix work,
        split-bf16 on the node rows: meets BASELINE configs[2]'s 1e-2) | 'bf16' (single rounding everywhere; ~2e-2
        on the object logits, outside that tolerance -- kept for comparison) | 'bf16x3_attn1' (split-bf16 everywhere except the
        edge cross-attention -- its three projections and the attention itself -- which is single-rounded: the 3D outputs never
        see that block and keep the split-bf16 accuracy, the 2D outputs hold 1e-2 on weights where 'bf16_mixed' does not,
        profiles/r05_probes/precision_mix_study.txt) | 'fp16_mixed' ('bf16_mixed' with fp16 instead of bf16 in the half-row tensors and on the
        matrix cores of the edge-row kernels -- the same MFMA rate, eight times finer rounding; values beyond +-65504 saturate).  Softmax/LN and HBM tensors stay fp32."""
        if mode not in self.PRECISIONS:
            raise L.VlsatError(f"gemm precision must be one of {sorted(self.PRECISIONS)}")
        L.check(self._lib.vlsat_set_gemm_precision(self._h, self.PRECISIONS[mode]))
        self.gemm_precision = mode
        self._drop_replicas()
        return self

    @torch.no_grad()
    def auto_precision(self, obj_points, obj_2d_feats, edge_indices, descriptor=None, batch_ids=None, tol: float = 1e-2,
                       margin: float = 0.5, candidates: Sequence[str] = ("bf16_mixed", "fp16_mixed", "bf16x3_attn1", "bf16x3")) -> dict:
        """Pick the fastest bf16 mode whose outputs stay inside ``tol`` on THIS checkpoint: one calibration batch is run in
        split-bf16 ('bf16x3', ~1e-5 from fp32 at Xavier scale and 1e-3 up to twice that) as the
        reference and in each faster candidate; the first candidate whose largest output difference is below ``margin * tol``
        is set (the margin covers batches the calibration did not see).  'bf16_mixed' holds BASELINE configs[2]'s 1e-2 on
        Xavier-scale weights with a 2x margin but not on a network whose weights amplify roundoff (LayerNorm gains above
        ~1.5, section 8): the error of the single-rounding modes is the rounding of the MFMA operands themselves (storing the
        edge tensors as hi/lo pairs instead of bf16 leaves it unchanged, measured), so the remedy is the mode, not a format:
        'fp16_mixed' (the same kernels on fp16 operands: 1/8 of the rounding, 3-4 % slower),
        then 'bf16x3_attn1', then split-bf16.
        Returns {'mode', 'errors': {candidate: max-abs difference}}.  Costs one forward per candidate plus one reference."""
        prev = self.gemm_precision
        self.set_gemm_precision("bf16x3")
        ref = [o.clone() for o in self.forward(obj_points, obj_2d_feats, edge_indices, descriptor, batch_ids)]
        errors, chosen = {}, "bf16x3"
        for mode in candidates:
            if mode == "bf16x3":
                errors[mode] = 0.0
                chosen = mode
                break
            self.set_gemm_precision(mode)
            got = self.forward(obj_points, obj_2d_feats, edge_indices, descriptor, batch_ids)
            errors[mode] = max(float((g - r).abs().max()) if g.numel() else 0.0 for g, r in zip(got, ref))
            if errors[mode] <= margin * tol:
                chosen = mode
                break
        else:
            chosen = "bf16x3" if "bf16x3" in candidates else prev
        self.set_gemm_precision(chosen)
        return {"mode": chosen, "errors": errors, "tol": tol, "margin": margin}

    BATCH_MODES = {"per_scene": 0, "reference": 1}

    def set_batch_mode(self, mode: str):
        """How a call that carries several scenes is evaluated.  'per_scene' (default): every scene exactly as if it
        were evaluated alone -- MMGNet.validation's contract (batch_size=1, reference src/model/model.py:185).
        'reference': what Mmgnet.forward itself computes on such a batch -- its edge cross-attention has no scene mask
        (network_MMG.py:228-234), so 2D edges attend to the 3D edges of every scene in the call."""
        if mode not in self.BATCH_MODES:
            raise L.VlsatError(f"batch mode must be one of {sorted(self.BATCH_MODES)}")
        L.check(self._lib.vlsat_set_edge_attention_scope(self._h, self.BATCH_MODES[mode]))
        self._drop_plans()                      # plans bake the attention tile table in
        self.batch_mode = mode
        self._drop_replicas()
        return self

    def debug_option(self, name: str, value: int):
        """Switches of the library handle (``vlsat_debug_option``; names and meanings: include/vlsat.h).  Defaults are the
        measured-best settings; none changes results beyond floating-point summation order / the mode's rounding.  Lab switches
        (timing ablations and the like) exist in the experiments build only (``build.py --experiments``)."""
        L.check(self._lib.vlsat_debug_option(self._h, name.encode(), int(value)))
        self._debug_options[name] = int(value)
        if name in ("dual_stream", "flash_split", "flash_bq_big_min"):
            self._drop_plans()
        self._drop_replicas()                   # (replicas built before this call run another kernel configuration)
        return self

    def _drop_plans(self):
        for p in self._plans.values():
            p.destroy()
        self._plans.clear()
        self._ident.clear()

    # ---- nn.Module-like surface --------------------------------------------------------------
    def eval(self):
        self.training = False
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise L.VlsatError("weights live on the device given at construction; build a new VLSATModel to move")
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def close(self):
        self._drop_replicas()
        self._drop_plans()
        if getattr(self, "_h", None):
            self._lib.vlsat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def load_state(self, weights: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        """``weights``: reference ``state_dict`` keys (``'mmg.gcn_3ds.0.edgeatten.nn_edge.0.weight'``)
        -> fp32 arrays.  Dead/unused entries of a reference checkpoint are ignored.  May be called again on a
        loaded model (like BaseModel.load): the previous device weights are dropped, plans stay valid."""
        want = param_shapes(self.config)
        missing = [k for k in want if k not in weights]
        if missing and strict:
            raise L.VlsatError(f"missing {len(missing)} weights, first: {missing[0]}")
        # every shape is checked BEFORE the first upload: the first vlsat_load_weight on a finalised handle frees the
        # device weights, so a reload that fails half way must not be started for a reason known in advance
        host = {}
        for k, shape in want.items():
            if k not in weights:
                continue
            v = weights[k]
            if torch.is_tensor(v):
                v = v.detach().cpu().numpy()
            v = np.ascontiguousarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(shape):
                raise L.VlsatError(f"weight {k}: shape {v.shape}, expected {shape}")
            host[k] = v
        self._loaded = False                    # until the finalize below succeeds, forward() refuses to run
        self._drop_replicas()                   # (they hold the previous weights)
        with torch.cuda.device(self.device):
            for k, v in host.items():
                L.check(self._lib.vlsat_load_weight(self._h, k.encode(), v.ctypes.data, v.size))
            L.check(self._lib.vlsat_finalize_weights(self._h))
        self._loaded = True
        self._host_weights = host               # (kept for replicate(): 136 MB of host memory at the default sizes)
        return self

    def replicate(self) -> "VLSATModel":
        """A second model with the same configuration, weights, precision, batch mode and debug options on the same GPU: its own library
        handle, plans and scratch, so it can be driven from another host thread on another stream at the same time
        (a handle serves one thread and one stream at a time, include/vlsat.h).  evaluate.validation(workers=K) uses K - 1."""
        if not self._loaded:
            raise L.VlsatError("replicate(): load weights first")
        m = VLSATModel(self.config, str(self.device)).load_state(self._host_weights)
        m._host_weights = self._host_weights
        if self.gemm_precision != "fp32":
            m.set_gemm_precision(self.gemm_precision)
        if self.batch_mode != "per_scene":
            m.set_batch_mode(self.batch_mode)
        for name, value in self._debug_options.items():      # same kernel configuration as this model (A/B runs with workers > 1)
            m.debug_option(name, value)
        m.training = self.training
        return m

    def replicas(self, k: int) -> list:
        """``k`` replicas (``replicate()``), built on first use and kept until ``close()`` / the next ``load_state`` /
        a precision or batch-mode change; evaluate.validation(workers=K) takes K - 1 of them on every call."""
        cur = getattr(self, "_replicas", None)
        if cur is None:
            cur = self._replicas = []
        while len(cur) < k:
            cur.append(self.replicate())
        return cur[:k]

    def _drop_replicas(self):
        for m in getattr(self, "_replicas", None) or []:
            m.close()
        self._replicas = []

    def load(self, ckpt_dir: str, best: bool = False) -> bool:
        """``BaseModel.load(best)`` on the reference's checkpoint directory (one ``.pth`` per sub-module,
        reference model_utils/model_base.py:75-129); sets ``iteration`` / ``eva_res`` like the reference."""
        from .checkpoint import load_reference_checkpoint
        weights, meta = load_reference_checkpoint(ckpt_dir, self.config, best=best)
        self.load_state(weights)
        self.iteration, self.eva_res = meta["iteration"], meta["eva_res"]
        return True

    def process_val(self, obj_points, obj_2d_feats, gt_cls, descriptor, gt_rel_cls, edge_indices, batch_ids=None,
                    with_log=False, use_triplet=False):
        """Same arguments and 10-tuple as ``Mmgnet.process_val`` (reference SGFN_MMG/model.py:458-480);
        forward and ranking both run on the GPU (``metrics.process_val``)."""
        from . import metrics
        return metrics.process_val(self, obj_points, obj_2d_feats, gt_cls, descriptor, gt_rel_cls, edge_indices,
                                   batch_ids, use_triplet=use_triplet)

    # ---- graph plan ----------------------------------------------------------------------------
    @staticmethod
    def _fc_host(sizes: Sequence[int]):
        """Host copy of the canonical fully-connected graph: source-major ordered pairs without self loops per scene,
        node offsets applied (reference dataset_3dssg.py:264-266 + collate_fn_mmg DataLoader.py:160-172)."""
        ei, bid, off = [], [], 0
        for s, n in enumerate(sizes):
            a = np.repeat(np.arange(n, dtype=np.int64), n)
            b = np.tile(np.arange(n, dtype=np.int64), n)
            keep = a != b
            ei.append(np.stack([a[keep], b[keep]], 0) + off)
            bid.append(np.full(n, s, dtype=np.int64))
            off += n
        return torch.from_numpy(np.ascontiguousarray(np.concatenate(ei, 1))), torch.from_numpy(np.concatenate(bid))

    def _plan(self, edge_indices, batch_ids, n, p, fc_sizes: Optional[Sequence[int]] = None) -> _Plan:
        """Plan for this graph, from a content-keyed LRU cache.  In order of cost:
          1. ``fc_sizes`` given: the caller states that edge_indices IS the canonical fully-connected edge list of
             scenes with these object counts (source-major, ``synth.fc_edges`` order) -> key (sizes, P).  The claim is
             checked for the FIRST tensor that arrives under a key only: host-side edge tensors against the canonical list
             on the host, device tensors (and device batch_ids) by a kernel against the new plan's own tables
             (``vlsat_plan_check_graph``: one 4-byte read-back when the plan is built; a wrongly ordered list would
             attribute every rel_cls row to the wrong edge).  Later calls with the same sizes are TRUSTED -- a key that is
             already cached reads nothing from the device -- unless ``self.verify_fc_every_call`` is set (a debugging aid:
             every call with the hint then re-runs the check, host or device, at the price of a stream synchronisation);
          2. the same tensor OBJECTS as an earlier call, unmodified -> no copy either;
          3. edge_indices / batch_ids on the host (the reference's loader yields them there) -> hashed on the host;
          4. device tensors never seen before -> one D2H copy (a stream sync) to hash them.
        In every case a graph with the same content re-uses its plan: no rebuild, no upload."""
        if batch_ids is None:
            batch_ids = self._zero_bid.get(n)
            if batch_ids is None:
                batch_ids = self._zero_bid[n] = torch.zeros(n, 1, dtype=torch.int64, device=self.device)
        if (not torch.is_tensor(batch_ids) or batch_ids.numel() != n or batch_ids.dtype != torch.int64
                or not torch.is_tensor(edge_indices) or edge_indices.dtype != torch.int64 or edge_indices.dim() != 2
                or edge_indices.shape[0] != 2):
            raise L.VlsatError(f"edge_indices must be int64[2,E] and batch_ids int64[{n},1] (or [{n}])")
        key, ei, bid = None, None, None
        if fc_sizes is not None:
            sizes = tuple(int(x) for x in fc_sizes)
            if sum(sizes) != n or edge_indices.shape[1] != sum(k * (k - 1) for k in sizes):
                raise L.VlsatError("fc_sizes does not match the node / edge counts")
            key = ("fc", sizes, p, self.batch_mode)
            if self.verify_fc_every_call and key in self._plans:         # (debugging aid: the claim of an already trusted key, again)
                if edge_indices.is_cuda or batch_ids.is_cuda:
                    self._check_device_graph(self._plans[key], edge_indices, batch_ids)
                elif not torch.equal(edge_indices.contiguous(), self._fc_host(sizes)[0]):
                    raise L.VlsatError("fc_sizes: edge_indices is not the canonical fully-connected edge list of these scenes")
            if not edge_indices.is_cuda and key not in self._plans:      # free to verify on the host, once per key
                if not torch.equal(edge_indices.contiguous(), self._fc_host(sizes)[0]):
                    raise L.VlsatError("fc_sizes: edge_indices is not the canonical fully-connected edge list of these scenes")
        else:
            ik = (id(edge_indices), id(batch_ids))
            hit = self._ident.get(ik)
            if hit is not None:
                refs, versions, k = hit
                if (refs[0]() is edge_indices and refs[1]() is batch_ids and k in self._plans
                        and versions == (edge_indices._version, batch_ids._version)):
                    self._plans.move_to_end(k)
                    self.plan_stats["identity_hits"] += 1
                    return self._plans[k]
                del self._ident[ik]
            if edge_indices.is_cuda or batch_ids.is_cuda:
                self.plan_stats["d2h_copies"] += 1
            ei = edge_indices.detach().cpu().contiguous()
            bid = batch_ids.detach().view(-1).cpu().contiguous()
            # scene ids only matter through the partition they induce: hash the run lengths, not the values -- after checking
            # that no id comes back in a later run ([0,1,0] has the cuts of [0,1,2] but is not a valid batch: an uncached call
            # fails with "nodes of a scene must be contiguous", and so must a cached one)
            cuts = torch.nonzero(bid[1:] != bid[:-1]).view(-1).numpy() if n > 1 else np.zeros(0, np.int64)
            if len(cuts):
                runs = bid[np.concatenate([[0], cuts + 1])].numpy()
                if len(np.unique(runs)) != len(runs):
                    raise L.VlsatError("batch_ids: the nodes of a scene must be contiguous (a scene id appears in two runs)")
            hsh = hashlib.blake2b(digest_size=16)
            hsh.update(np.ascontiguousarray(cuts).tobytes())
            hsh.update(ei.numpy().tobytes())
            key = ("g", n, ei.shape[1], p, self.batch_mode, hsh.digest())
        plan = self._plans.get(key)
        if plan is not None:
            self._plans.move_to_end(key)
            self.plan_stats["hits"] += 1
        else:
            if ei is None:
                ei, bid = self._fc_host(sizes)
            plan = self._build(ei, bid, n, p)
            if fc_sizes is not None and (edge_indices.is_cuda or batch_ids.is_cuda) and key not in self._fc_verified:
                try:
                    self._check_device_graph(plan, edge_indices, batch_ids)
                except L.VlsatError:
                    plan.destroy()
                    raise
                if len(self._fc_verified) > 8192:
                    self._fc_verified.clear()
                self._fc_verified.add(key)         # (a key checked once stays checked when its plan is evicted and rebuilt)
            self._plans[key] = plan
            self.plan_stats["builds"] += 1
            total = sum(q.ws_bytes for q in self._plans.values())
            while len(self._plans) > 1 and (len(self._plans) > self.MAX_PLANS or total > self.MAX_PLAN_BYTES):
                _, old = self._plans.popitem(last=False)
                total -= old.ws_bytes
                old.destroy()
        if fc_sizes is None:
            self._ident[(id(edge_indices), id(batch_ids))] = (
                (weakref.ref(edge_indices), weakref.ref(batch_ids)), (edge_indices._version, batch_ids._version), key)
            while len(self._ident) > 4 * self.MAX_PLANS:
                self._ident.popitem(last=False)
        return plan

    def _check_device_graph(self, plan: "_Plan", edge_indices, batch_ids):
        """fc_sizes with device tensors, first use of the key: the device edge list / batch ids against the plan's tables."""
        if plan.perm is not None:
            raise L.VlsatError("fc_sizes: the canonical edge list is grouped by scene; this plan is not")
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        ei_d = edge_indices.to(self.device).contiguous()
        bid_d = batch_ids.to(self.device).view(-1).contiguous()
        L.check(self._lib.vlsat_plan_check_graph(plan.handle, L.ptr(ei_d), L.ptr(bid_d), L.ptr(flag), L.stream_ptr()))
        bad = int(flag.item())
        if bad:
            raise L.VlsatError(f"fc_sizes: edge_indices / batch_ids on the device are not the canonical fully-connected graph of "
                               f"these scenes ({bad} mismatching entries)")

    def _build(self, ei: torch.Tensor, bid: torch.Tensor, n: int, p: int) -> _Plan:
        e = ei.shape[1]
        perm = None
        out = C.c_void_p()

        def create(ei_host):
            return self._lib.vlsat_plan_create(self._h, bid.data_ptr(), ei_host.data_ptr(), n, e, p, C.byref(out))

        rc = create(ei)
        if rc == -4:   # VLSAT_EGRAPH: edges not grouped by scene -> stable sort by scene, remember the permutation
            order = torch.argsort(bid[ei[0]], stable=True)
            ei = ei[:, order].contiguous()
            rc = create(ei)
            perm = order.to(self.device)
        L.check(rc)
        ws = C.c_size_t()
        L.check(self._lib.vlsat_plan_info(out, None, C.byref(ws), None))
        return _Plan(out, perm, ws.value)

    def plan_info(self, edge_indices, batch_ids, n, p):
        plan = self._plan(edge_indices, batch_ids, n, p)
        s, ws, fc = C.c_int32(), C.c_size_t(), C.c_int32()
        L.check(self._lib.vlsat_plan_info(plan.handle, C.byref(s), C.byref(ws), C.byref(fc)))
        return {"n_scenes": s.value, "workspace_bytes": ws.value, "is_fc": bool(fc.value)}

    # ---- forward -------------------------------------------------------------------------------
    def _chk(self, t, name, shape, dtype):
        if not torch.is_tensor(t) or t.device != self.device:
            raise L.VlsatError(f"{name}: expected a tensor on {self.device}")
        if t.dtype != dtype:
            raise L.VlsatError(f"{name}: dtype {t.dtype}, expected {dtype}")
        for got, want in zip(t.shape, shape):
            if want is not None and got != want:
                raise L.VlsatError(f"{name}: shape {tuple(t.shape)}, expected {shape}")
        if len(t.shape) != len(shape):
            raise L.VlsatError(f"{name}: rank {t.dim()}, expected {len(shape)}")
        return t if t.is_contiguous() else t.contiguous()

    def _inputs(self, obj_points, obj_2d_feats, edge_indices, descriptor, need_2d=True):
        if not self._loaded:
            raise L.VlsatError("weights not loaded: call load_state() first")
        if descriptor is None:
            raise L.VlsatError("descriptor is required (MODEL.USE_SPATIAL must be true, SURVEY §8a)")
        c = self.config
        n = obj_points.shape[0]
        pts = self._chk(obj_points, "obj_points", (n, c.dim_point, None), torch.float32)
        f2d = self._chk(obj_2d_feats, "obj_2d_feats", (n, c.clip_feat_dim), torch.float32) if need_2d else None
        desc = self._chk(descriptor, "descriptor", (n, c.dim_descriptor), torch.float32)
        if not torch.is_tensor(edge_indices) or edge_indices.dim() != 2 or edge_indices.shape[0] != 2:
            raise L.VlsatError("edge_indices: expected int64[2,E]")
        return pts, f2d, desc, n, pts.shape[2], edge_indices.shape[1]

    @torch.no_grad()
    def forward(self, obj_points, obj_2d_feats, edge_indices, descriptor=None, batch_ids=None, istrain=False,
                fc_sizes: Optional[Sequence[int]] = None):
        """Same contract as ``Mmgnet.forward`` (reference SGFN_MMG/model.py:288-335):
        obj_points f32[N,3,P], obj_2d_feats f32[N,512], edge_indices i64[2,E], descriptor f32[N,11],
        batch_ids i64[N,1] -> (obj_logits_3d [N,160], obj_logits_2d [N,160], rel_cls_3d [E,26], rel_cls_2d [E,26]).
        ``edge_indices`` / ``batch_ids`` may live on the host or on the device (they only feed the graph plan, see
        ``_plan``); ``fc_sizes`` (objects per scene) optionally declares the canonical fully-connected graph.
        ``istrain=True`` returns the reference's 8-tuple (:332-333) -- forward only, modules in eval mode, no autograd."""
        pts, f2d, desc, n, p, e = self._inputs(obj_points, obj_2d_feats, edge_indices, descriptor)
        c = self.config
        with torch.cuda.device(self.device):
            plan = self._plan(edge_indices, batch_ids, n, p, fc_sizes)
            obj3 = torch.empty(n, c.num_obj_class, dtype=torch.float32, device=self.device)
            obj2 = torch.empty_like(obj3)
            rel3 = torch.empty(e, c.num_rel_class, dtype=torch.float32, device=self.device)
            rel2 = torch.empty_like(rel3)
            extras = ()
            if istrain:
                if not c.train_outputs:
                    raise L.VlsatError("forward(istrain=True) needs VLSATConfig(train_outputs=True) and the "
                                       "triplet_projector_2d weights")
                m3 = torch.empty(n, 512, dtype=torch.float32, device=self.device)
                m2 = torch.empty_like(m3)
                dis = torch.empty(e, 512, dtype=torch.float32, device=self.device)
                L.check(self._lib.vlsat_forward_train(self._h, plan.handle, pts.data_ptr(), f2d.data_ptr(), desc.data_ptr(),
                                                      obj3.data_ptr(), obj2.data_ptr(), rel3.data_ptr(), rel2.data_ptr(),
                                                      m3.data_ptr(), m2.data_ptr(), dis.data_ptr(), L.stream_ptr()))
                extras = (m3, m2, dis)
            else:
                L.check(self._lib.vlsat_forward(self._h, plan.handle, pts.data_ptr(), f2d.data_ptr(), desc.data_ptr(),
                                                obj3.data_ptr(), obj2.data_ptr(), rel3.data_ptr(), rel2.data_ptr(),
                                                L.stream_ptr()))
            if plan.perm is not None:     # outputs were computed in scene-grouped edge order
                def unperm(x):
                    y = torch.empty_like(x)
                    y[plan.perm] = x
                    return y
                rel3, rel2 = unperm(rel3), unperm(rel2)
                if istrain:
                    extras = (extras[0], extras[1], unperm(extras[2]))
        if istrain:
            scale = torch.tensor(math.exp(c.obj_logit_scale), dtype=torch.float32, device=self.device)
            return (obj3, obj2, rel3, rel2) + extras + (scale,)
        return obj3, obj2, rel3, rel2

    @torch.no_grad()
    def process_val_counts(self, counts, obj_points, obj_2d_feats, gt_cls, descriptor, gt_rel_multihot, edges_e2, batch_ids=None,
                           n_scenes: int = 1, fc_sizes: Optional[Sequence[int]] = None) -> bool:
        """``vlsat_process_val_counts``: forward + ranking of both branches + the additive counts of ``evaluate.fields()`` for one
        batch in ONE library call, intermediates in the plan's scratch (nothing allocated, nothing read back).  ``edges_e2`` is the
        loader's int64 [E,2] list, ``gt_rel_multihot`` int64 [E,R], ``gt_cls`` int64 [N], ``counts`` the device int64 vector.
        Returns False (nothing enqueued) when the plan had to permute the edges -- the caller then takes the separate calls."""
        ei = edges_e2.t() if fc_sizes is not None else edges_e2.t().contiguous()
        pts, f2d, desc, n, p, e = self._inputs(obj_points, obj_2d_feats, ei, descriptor)
        with torch.cuda.device(self.device):
            plan = self._plan(ei, batch_ids, n, p, fc_sizes)
            if plan.perm is not None:
                return False
            L.check(self._lib.vlsat_process_val_counts(self._h, plan.handle, pts.data_ptr(), f2d.data_ptr(), desc.data_ptr(),
                                                       gt_cls.data_ptr(), gt_rel_multihot.data_ptr(), edges_e2.data_ptr(), int(n_scenes),
                                                       counts.data_ptr(), L.stream_ptr()))
        return True

    @torch.no_grad()
    def forward_3d(self, obj_points, edge_indices, descriptor, batch_ids=None, fc_sizes: Optional[Sequence[int]] = None):
        """3D-only deployment (no image features): returns (obj_logits_3d, rel_cls_3d), bit-identical to
        the first and third outputs of ``forward`` -- the 3D branch never reads the 2D branch
        (cf. reference src/model/SGFN_MMG/model_single.py:247-281) -- at about half the work."""
        pts, _, desc, n, p, e = self._inputs(obj_points, None, edge_indices, descriptor, need_2d=False)
        c = self.config
        with torch.cuda.device(self.device):
            plan = self._plan(edge_indices, batch_ids, n, p, fc_sizes)
            obj3 = torch.empty(n, c.num_obj_class, dtype=torch.float32, device=self.device)
            rel3 = torch.empty(e, c.num_rel_class, dtype=torch.float32, device=self.device)
            L.check(self._lib.vlsat_forward(self._h, plan.handle, pts.data_ptr(), None, desc.data_ptr(),
                                            obj3.data_ptr(), None, rel3.data_ptr(), None, L.stream_ptr()))
            if plan.perm is not None:
                r3 = torch.empty_like(rel3)
                r3[plan.perm] = rel3
                rel3 = r3
        return obj3, rel3

    # ---- profiling / debug hooks -----------------------------------------------------------------
    def profile_enable(self, on: bool):
        L.check(self._lib.vlsat_profile_enable(self._h, int(on)))

    def profile_read(self) -> dict:
        out = {}
        for i in range(self._lib.vlsat_profile_num_classes()):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            L.check(self._lib.vlsat_profile_read(self._h, i, C.byref(ms), C.byref(n), C.byref(fl)))
            out[self._lib.vlsat_profile_class_name(i).decode()] = {"ms": ms.value, "launches": n.value,
                                                                  "flops": fl.value}
        return out

    def debug_stop_after(self, stage: int):
        L.check(self._lib.vlsat_debug_stop_after(self._h, stage))

    def debug_buffer(self, edge_indices, batch_ids, n, p, name: str) -> torch.Tensor:
        """Copy of a named workspace buffer of the plan for this graph (tests only)."""
        plan = self._plan(edge_indices, batch_ids, n, p)
        ptr, rows, cols, ld = C.c_void_p(), C.c_int64(), C.c_int32(), C.c_int32()
        L.check(self._lib.vlsat_debug_buffer(plan.handle, name.encode(), C.byref(ptr), C.byref(rows), C.byref(cols),
                                             C.byref(ld)))
        out = torch.empty(rows.value, cols.value, dtype=torch.float32, device=self.device)
        L.check(self._lib.vlsat_debug_read(plan.handle, name.encode(), out.data_ptr(), cols.value))
        return out
