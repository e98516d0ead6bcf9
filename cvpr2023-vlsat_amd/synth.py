"""Deterministic synthetic weights and 3RScan-shaped scenes (numpy only).

Both generators are pure functions of (name/seed) so that this container (where the
golden vectors are made from the real reference) and the GPU box (which never sees the
reference) regenerate bit-identical tensors without shipping >100 MB of fixtures.

Scene semantics restate the reference input contract:
  * per-object descriptor = ``gen_descriptor`` on the raw sampled points
    (reference ``src/utils/op_utils.py:47-64``: mean, unbiased std, max-min, volume, max dim),
  * points are then zero-meaned per object (reference ``src/dataset/dataset_3dssg.py:291-293``),
  * fully-connected directed edges, source-major, no self loops
    (reference ``src/dataset/dataset_3dssg.py:264-266``),
  * a batch is the concatenation of scenes with node-index offsets and ``batch_ids [N,1]``
    (reference ``src/dataset/DataLoader.py:153-176``),
  * ``obj_points`` is handed to the model as ``[N,3,P]`` (reference ``src/model/model.py:79``).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

from .config import VLSATConfig, param_shapes


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(name.encode())])


def make_weights(cfg: VLSATConfig, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Formula weights: xavier-uniform matrices, small non-zero biases and deliberately
    non-trivial LayerNorm / BatchNorm statistics (default init would hide BN-folding and
    LN-affine bugs)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        g = _rng(name, seed)
        leaf = name.rsplit(".", 1)[1]
        is_norm = (".layer_norm." in name or name.startswith("mlp_3d.1.") or any(f".bn{i}." in name for i in range(1, 6))
                   or name.startswith("mmg.self_attn_fc.2.") or name.startswith("mmg.self_attn_fc.5."))
        if is_norm:
            if leaf == "weight":
                w = g.uniform(0.9, 1.1, shape)
            elif leaf == "bias":
                w = g.uniform(-0.05, 0.05, shape)
            elif leaf == "running_mean":
                w = g.uniform(-0.1, 0.1, shape)
            else:  # running_var
                w = g.uniform(0.5, 1.5, shape)
        elif leaf == "bias":
            w = g.uniform(-0.05, 0.05, shape)
        else:
            fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
            a = np.sqrt(6.0 / (fan_in + fan_out))
            w = g.uniform(-a, a, shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def make_weights_stress(cfg: VLSATConfig, scale: float, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Trained-scale stress weights: the formula weights with every GCN matrix (nn_edge, proj_*, the gate MLP, prop)
    multiplied by ``scale`` and LayerNorm gains drawn from U(0.3, 3).  Exercises the node-side hoisting of
    nn_edge.0 / proj_query / gate layer 1 (reference network_MMG.py:84-112) away from Xavier scale, where a
    summation-order slip would hide below the tolerance."""
    w = make_weights(cfg, seed)
    for name in list(w):
        if ".layer_norm.weight" in name:
            w[name] = _rng(name + "/stress", seed).uniform(0.3, 3.0, w[name].shape).astype(np.float32)
        elif (".gcn_2ds." in name or ".gcn_3ds." in name) and name.endswith(".weight"):
            w[name] = (w[name] * np.float32(scale)).astype(np.float32)
    return w


def fc_edges(n: int) -> np.ndarray:
    """[2,E] source-major fully-connected pairs without self loops, E = n(n-1)."""
    src = np.repeat(np.arange(n, dtype=np.int64), n)
    dst = np.tile(np.arange(n, dtype=np.int64), n)
    keep = src != dst
    return np.stack([src[keep], dst[keep]], 0)


def gen_descriptor(raw: np.ndarray) -> np.ndarray:
    """raw [N,P,3] float32 -> [N,11] (centroid, unbiased std, dims, volume, max dim)."""
    raw64 = raw.astype(np.float64)
    mean = raw64.mean(1)
    std = raw64.std(1, ddof=1)
    dims = raw64.max(1) - raw64.min(1)
    vol = dims.prod(-1, keepdims=True)
    length = dims.max(-1, keepdims=True)
    return np.concatenate([mean, std, dims, vol, length], -1).astype(np.float32)


def make_scene(n_obj: int, n_pts: int, seed: int, clip_dim: int = 512) -> dict:
    """One scene: dict of numpy arrays in the reference's model-input layout."""
    g = np.random.default_rng([int(seed) & 0x7FFFFFFF, 0x3D55])
    centre = g.uniform(0.0, 4.0, (n_obj, 1, 3))
    ext = g.uniform(0.2, 1.0, (n_obj, 1, 3))
    raw = (centre + g.uniform(-0.5, 0.5, (n_obj, n_pts, 3)) * ext).astype(np.float32)
    desc = gen_descriptor(raw)
    pts = raw - raw.mean(1, keepdims=True, dtype=np.float64).astype(np.float32)
    f2d = g.standard_normal((n_obj, clip_dim))
    f2d = (f2d / np.linalg.norm(f2d, axis=-1, keepdims=True)).astype(np.float32)
    return {
        "obj_points": np.ascontiguousarray(pts.transpose(0, 2, 1)),   # [N,3,P]
        "obj_2d_feats": f2d,                                           # [N,512]
        "edge_indices": fc_edges(n_obj),                               # [2,E] int64
        "descriptor": desc,                                            # [N,11]
        "batch_ids": np.zeros((n_obj, 1), dtype=np.int64),            # [N,1]
    }


def collate(scenes: list) -> dict:
    """Concatenate scenes the way ``collate_fn_mmg`` does (reference DataLoader.py:153-176):
    node tensors stacked, edge indices offset by the running node count, batch_ids = scene id."""
    off, ei, bid = 0, [], []
    for s, sc in enumerate(scenes):
        n = sc["obj_points"].shape[0]
        ei.append(sc["edge_indices"] + off)
        bid.append(np.full((n, 1), s, dtype=np.int64))
        off += n
    return {
        "obj_points": np.concatenate([s["obj_points"] for s in scenes], 0),
        "obj_2d_feats": np.concatenate([s["obj_2d_feats"] for s in scenes], 0),
        "edge_indices": np.concatenate(ei, 1),
        "descriptor": np.concatenate([s["descriptor"] for s in scenes], 0),
        "batch_ids": np.concatenate(bid, 0),
    }


def make_batch(n_scenes: int, n_obj: int, n_pts: int, seed0: int = 1000) -> dict:
    """Batch of ``n_scenes`` scenes with seeds seed0, seed0+1, ...."""
    return collate([make_scene(n_obj, n_pts, seed0 + s) for s in range(n_scenes)])


# ---- config-switch parity cases (tests/golden/make_golden_switches.py makes their goldens from the real reference) ----
SWITCH_CASES = {
    "switch_with_bn": dict(N_LAYERS=2, WITH_BN=True),
    "switch_no_gcn_edge": dict(N_LAYERS=2, USE_GCN_EDGE=False),
    "switch_single_rel": dict(N_LAYERS=2, multi_rel_outputs=False, num_rel_class=27),
    "switch_rgb_normal": dict(N_LAYERS=2, USE_RGB=True, USE_NORMAL=True),
    "switch_feature_transform": dict(N_LAYERS=1, feature_transform=True),
    "switch_all": dict(feature_transform=True, N_LAYERS=1, GCN_AGGR="mean", WITH_BN=True, USE_GCN_EDGE=False, multi_rel_outputs=False,
                       num_rel_class=27, USE_RGB=True, USE_NORMAL=True),
}


def switch_scenes(cfg: VLSATConfig) -> list:
    """The ragged pair of scenes (4 and 6 objects x 32 points) every switch case runs on; colour / normal channels
    (cfg.dim_point > 3) are seeded uniform values in [-1, 1] appended after xyz."""
    out = []
    for i, n in enumerate((4, 6)):
        sc = make_scene(n, 32, 5000 + i)
        if cfg.dim_point > 3:
            g = np.random.default_rng([5000 + i, cfg.dim_point])
            extra = g.uniform(-1, 1, (n, cfg.dim_point - 3, 32)).astype(np.float32)
            sc["obj_points"] = np.concatenate([sc["obj_points"], extra], 1)
        out.append(sc)
    return out
