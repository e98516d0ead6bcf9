"""Reader / writer for the reference's checkpoint directory layout.

``BaseModel.save`` (reference ``src/model/model_utils/model_base.py:47-73``) writes, under
``<PATH>/ckp/Mmgnet/<exp>/``, ONE file per top-level sub-module -- ``obj_encoder.pth``,
``mmg.pth``, ``rel_predictor_3d.pth`` ... (``_best.pth`` suffix for the best model) -- each holding
``{'model': sub_module.state_dict()}`` (``saveWeights`` :150-158; keys may carry a DataParallel
``module.`` prefix, ``loadWeights`` :160-181), plus ``config{,_best}.pth`` = ``{'iteration', 'eva_res'}``.
``obj_logit_scale`` is a bare ``nn.Parameter`` and is never written.

``load_reference_checkpoint`` returns the flat ``{'<module>.<key>': array}`` dict that
``VLSATModel.load_state`` takes.  Only torch.load on the CPU is used here (glue, no arithmetic).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

from .config import VLSATConfig, param_shapes

BEST, LAST = "_best.pth", ".pth"


def _modules(cfg: VLSATConfig):
    seen = OrderedDict()
    for k in param_shapes(cfg):
        seen[k.split(".", 1)[0]] = True
    return list(seen)


def pick_suffix(ckpt_dir: str, best: bool) -> str:
    """Suffix selection of BaseModel.load (model_base.py:79-101): the best model when asked for or
    when it is the only one; otherwise whichever of checkpoint / best has the larger iteration."""
    cfg_best, cfg_last = os.path.join(ckpt_dir, "config" + BEST), os.path.join(ckpt_dir, "config" + LAST)
    if best:
        return BEST
    has_b, has_l = os.path.exists(cfg_best), os.path.exists(cfg_last)
    if has_b and not has_l:
        return BEST
    if has_b and has_l:
        it_l = torch.load(cfg_last, map_location="cpu", weights_only=False).get("iteration", 0)
        it_b = torch.load(cfg_best, map_location="cpu", weights_only=False).get("iteration", 0)
        return LAST if it_l > it_b else BEST
    if has_l:
        return LAST
    raise FileNotFoundError(f"no saved model under {ckpt_dir}")


def load_reference_checkpoint(ckpt_dir: str, cfg: VLSATConfig, best: bool = True) -> Tuple[Dict[str, np.ndarray], dict]:
    """-> (weights for VLSATModel.load_state, {'iteration', 'eva_res', 'suffix'})."""
    suffix = pick_suffix(ckpt_dir, best)
    want = param_shapes(cfg)
    out: Dict[str, np.ndarray] = {}
    for mod in _modules(cfg):
        path = os.path.join(ckpt_dir, mod + suffix)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: the reference writes one file per sub-module; '{mod}' is missing")
        sd = torch.load(path, map_location="cpu", weights_only=False)["model"]
        for k, v in sd.items():
            if k.startswith("module."):                    # saved from nn.DataParallel
                k = k[7:]
            full = f"{mod}.{k}"
            if full in want:
                a = v.detach().to(torch.float32).cpu().numpy()
                if tuple(a.shape) != tuple(want[full]):
                    raise ValueError(f"{full}: shape {a.shape} in checkpoint, expected {want[full]}")
                out[full] = np.ascontiguousarray(a)
    missing = [k for k in want if k not in out]
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} tensors, first: {missing[0]}")
    meta = {"iteration": 0, "eva_res": 0, "suffix": suffix}
    cpath = os.path.join(ckpt_dir, "config" + suffix)
    if os.path.exists(cpath):
        c = torch.load(cpath, map_location="cpu", weights_only=False)
        meta.update(iteration=c.get("iteration", 0), eva_res=c.get("eva_res", 0))
    return out, meta


def save_reference_checkpoint(ckpt_dir: str, weights: Dict[str, np.ndarray], best: bool = True, iteration: int = 0,
                              eva_res: float = 0.0, data_parallel: bool = False) -> None:
    """Write ``weights`` in the reference layout (what BaseModel.save produces for these modules)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    suffix = BEST if best else LAST
    by_mod: Dict[str, OrderedDict] = OrderedDict()
    for k, v in weights.items():
        mod, rest = k.split(".", 1)
        by_mod.setdefault(mod, OrderedDict())[("module." if data_parallel else "") + rest] = torch.from_numpy(np.asarray(v))
    for mod, sd in by_mod.items():
        torch.save({"model": sd}, os.path.join(ckpt_dir, mod + suffix))
    torch.save({"iteration": iteration, "eva_res": eva_res}, os.path.join(ckpt_dir, "config" + suffix))
