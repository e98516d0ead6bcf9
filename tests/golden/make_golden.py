#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference
(/root/reference, read-only) on CPU in this container.

Run once here:  python tests/golden/make_golden.py
The GPU box never sees /root/reference; it only sees the .npz files this writes.

Nothing from the reference is copied: it is imported.  Third-party modules it needs but
the image lacks are replaced by minimal stand-ins that restate their *documented*
semantics:
  * torch_geometric.nn.conv.MessagePassing  -- only the private collect/aggregate hooks the
    reference calls (network_util.py:56-58,68-72; op_utils.py:73-76): x_i/x_j gather by
    flow direction and scatter add/mean/max with empty segment -> 0;
  * src.lib.pointnet.graph, tkinter, clip  -- imported by the reference but unused on the
    eval arithmetic path;
  * Tensor.cuda -> identity (network_MMG.py:185-186 hard-codes .cuda());
  * Mmgnet.get_label_weight -> fixed tensors (it only sets *initial* classifier weights,
    which we overwrite with the formula weights anyway).
Inputs and weights come from vlsat_amd.synth (seeded formulas), so only outputs and a few
intermediate taps are stored.
"""
import inspect
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import vlsat_amd  # noqa: E402
from vlsat_amd import VLSATConfig, synth  # noqa: E402


# ----------------------------------------------------------------------------- stand-ins
class _Inspector:
    def __init__(self, owner):
        self.owner = owner

    def distribute(self, name, data):
        params = list(inspect.signature(getattr(self.owner, name)).parameters)
        if name == "aggregate":
            params = params[1:]
        return {k: data[k] for k in params if k in data}


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim
        self.inspector = _Inspector(self)
        self.__user_args__ = None

    def __check_input__(self, edge_index, size):
        return [None, None]

    def __collect__(self, args, edge_index, size, kwargs):
        i, j = (1, 0) if self.flow == "source_to_target" else (0, 1)
        out = {}
        for k, v in kwargs.items():
            out[k + "_i"] = v.index_select(0, edge_index[i])
            out[k + "_j"] = v.index_select(0, edge_index[j])
        out.update(index=edge_index[i], ptr=None, dim_size=None)
        return out

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        red = {"add": "sum", "mean": "mean", "max": "amax"}[self.aggr]
        base = torch.zeros(dim_size, inputs.shape[1], dtype=inputs.dtype)
        return base.scatter_reduce(0, index[:, None].expand_as(inputs), inputs,
                                   reduce=red, include_self=False)


def install_standins():
    for name in ("torch_geometric", "torch_geometric.nn", "torch_geometric.nn.conv"):
        m = types.ModuleType(name)
        m.MessagePassing = MessagePassing
        sys.modules[name] = m
    sys.modules["torch_geometric"].nn = sys.modules["torch_geometric.nn"]
    sys.modules["torch_geometric.nn"].conv = sys.modules["torch_geometric.nn.conv"]
    for name in ("src.lib", "src.lib.pointnet", "src.lib.pointnet.graph"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["src.lib.pointnet.graph"].GraphTripleConvNet = object
    tk = types.ModuleType("tkinter")
    tk.N = "n"
    sys.modules["tkinter"] = tk
    clip = types.ModuleType("clip")
    clip.clip = clip
    clip.tokenize = lambda *a, **k: None
    clip.load = lambda *a, **k: (None, None)
    sys.modules["clip"] = clip
    sys.modules["clip.clip"] = clip
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path[:0] = [REF, os.path.join(REF, "src")]


def build_reference(n_layers, aggr="max", num_rel=26, **model_overrides):
    from src.model.SGFN_MMG.model import Mmgnet
    from src.utils.config import Config

    cfg = Config(os.path.join(REF, "config", "mmgnet.json"))
    cfg.PATH = tempfile.mkdtemp(prefix="vlsat_golden_")
    cfg.exp = "golden"
    cfg.MODE = "eval"
    cfg.max_iteration = 10
    cfg.MODEL.N_LAYERS = n_layers
    cfg.MODEL.GCN_AGGR = aggr
    for k, v in model_overrides.items():          # USE_GCN_EDGE, WITH_BN, multi_rel_outputs, USE_RGB, USE_NORMAL, feature_transform
        setattr(cfg.MODEL, k, v)
    cfg.MODEL.adapter_path = os.path.join(REF, "clip_adapter", "checkpoint", "origin_mean.pth")

    def fixed_label_weight(self, *a, **k):
        g = torch.Generator().manual_seed(7)
        o = torch.randn(160, 512, generator=g)
        r = torch.randn(num_rel, 512, generator=g)
        return o / o.norm(dim=-1, keepdim=True), r / r.norm(dim=-1, keepdim=True)

    Mmgnet.get_label_weight = fixed_label_weight
    model = Mmgnet(cfg, 160, num_rel).eval()
    return model


def load_formula_weights(model, vcfg, seed=0):
    w = synth.make_weights(vcfg, seed)
    sd = model.state_dict()
    # dead at eval: triplet projectors, logit scales that are never checkpointed, BN counters, the BatchNorm
    # layers PointNetfeat creates under WITH_BN but whose output it discards (network_PointNet.py:141-164), and
    # proj_edge under USE_GCN_EDGE=False is still live in the state_dict and in our inventory (computed, unused)
    dead = ("triplet_projector_", "clip_adapter.obj_logit_scale", "obj_logit_scale", "num_batches_tracked",
            "obj_encoder.bn", "_encoder_2d.bn", "_encoder_3d.bn")
    live = [k for k in sd if not any(d in k for d in dead)]
    assert sorted(live) == sorted(w.keys()), (set(live) ^ set(w.keys()))
    for k, v in w.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k].copy_(torch.from_numpy(v))
    model.load_state_dict(sd)
    return w


def t(batch):
    return {k: torch.from_numpy(v) for k, v in batch.items()}


def run(model, b, taps=None):
    b = t(b)
    hooks, got = [], {}
    if taps:
        for name, key in taps.items():
            mod = dict(model.named_modules())[name]

            def mk(key):
                def hook(m, i, o):
                    if isinstance(o, tuple):
                        for n, x in enumerate(o):
                            if torch.is_tensor(x):
                                got[f"{key}.{n}"] = x.detach().clone().numpy()
                    else:
                        got[key] = o.detach().clone().numpy()
                return hook
            hooks.append(mod.register_forward_hook(mk(key)))
    with torch.no_grad():
        o = model(b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"],
                  istrain=False)
    for h in hooks:
        h.remove()
    res = dict(obj3d=o[0].numpy(), obj2d=o[1].numpy(), rel3d=o[2].numpy(), rel2d=o[3].numpy())
    res.update({"tap." + k: v for k, v in got.items()})
    return res


TAPS = {
    "obj_encoder": "obj_encoder",
    "mlp_3d": "mlp_3d",
    "rel_encoder_3d": "rel_encoder_3d",
    "rel_encoder_2d": "rel_encoder_2d",
    "clip_adapter": "clip_adapter",
    "mmg.self_attn_fc": "dist_bias",            # [1,n,n,8] for a single scene
    "mmg.self_attn.0": "self_attn0",
    "mmg.cross_attn.0": "cross_attn0",
    "mmg.gcn_3ds.0.edgeatten": "edgeatten3d0",  # .0 gated x [E,256], .1 edge' [E,512], .2 prob [E,32,8]
    "mmg.gcn_3ds.0": "gcn3d0",                  # .0 node' [N,512], .1 edge'
    "mmg.gcn_2ds.0": "gcn2d0",
    "mmg.cross_attn_rel.0": "cross_attn_rel0",
    "mmg": "mmg",                               # .0 node3d .1 node2d .2 edge3d .3 edge2d
}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    install_standins()
    meta = {}

    # ---- KAT: the reference's only executable example of the gather/scatter conventions
    #      (network_util.py:75-99), evaluated through the reference classes themselves.
    from src.model.model_utils.network_util import Aggre_Index, Gen_Index
    from src.utils.op_utils import Gen_edge_descriptor
    x = torch.zeros(3, 5)
    x[1] = 1
    x[2] = 2
    ei = torch.tensor([[0, 1, 2], [2, 1, 0]])
    kat = {}
    for flow in ("source_to_target", "target_to_source"):
        xi, xj = Gen_Index(flow=flow)(x, ei)
        kat[f"gen_index.{flow}.x_i"] = xi.numpy()
        kat[f"gen_index.{flow}.x_j"] = xj.numpy()
        tmp = -torch.arange(5, dtype=torch.float32)[:, None].repeat(1, 2)
        ei2 = torch.tensor([[0, 1, 2, 1, 0], [2, 1, 1, 1, 1]])
        for aggr in ("max", "add", "mean"):
            kat[f"aggre_index.{flow}.{aggr}"] = Aggre_Index(flow=flow, aggr=aggr)(tmp, ei2, dim_size=3).numpy()
    np.savez(os.path.join(HERE, "kat_index.npz"), **kat)

    # ---- cfg 1: one scene, 8 objects x 256 pts, L=2 (BASELINE.json configs[0]); with taps
    c1 = VLSATConfig(N_LAYERS=2)
    m = build_reference(2)
    load_formula_weights(m, c1)
    b = synth.make_batch(1, 8, 256, seed0=1000)
    r = run(m, b, TAPS)
    ed = Gen_edge_descriptor(flow="target_to_source")(torch.from_numpy(b["descriptor"]),
                                                      torch.from_numpy(b["edge_indices"]))
    r["tap.edge_descriptor"] = ed.squeeze(-1).numpy()
    np.savez_compressed(os.path.join(HERE, "cfg1_n8_p256_l2.npz"), **r)
    meta["cfg1"] = {k: v.shape for k, v in r.items()}

    # ---- ragged 2-scene batch (5 and 7 objects, 64 pts), L=2: per-scene reference calls
    #      (= validation()'s batch_size=1 contract) concatenated, plus the real
    #      batched call whose 3D branch is batch-independent.
    scenes = [synth.make_scene(5, 64, 2000), synth.make_scene(7, 64, 2001)]
    per = [run(m, synth.collate([s])) for s in scenes]
    cat = {k: np.concatenate([p[k] for p in per], 0) for k in ("obj3d", "obj2d", "rel3d", "rel2d")}
    bb = run(m, synth.collate(scenes))
    cat["batched_obj3d"] = bb["obj3d"]
    cat["batched_rel3d"] = bb["rel3d"]
    cat["batched_obj2d"] = bb["obj2d"]          # differ from the per-scene ones: the edge cross-attention of a
    cat["batched_rel2d"] = bb["rel2d"]          # multi-scene call is not masked per scene
    np.savez(os.path.join(HERE, "ragged_n5_n7_p64_l2.npz"), **cat)

    # ---- general (non fully-connected, unsorted, with empty source segments) edge list,
    #      all three aggregators, 6 objects x 32 pts, L=1 (depth==1 applies the ReLU, MMG :236)
    g = np.random.default_rng(5)
    sc = synth.make_scene(6, 32, 3000)
    allp = sc["edge_indices"]
    pick = g.permutation(allp.shape[1])[:13]
    keep = allp[:, pick]
    keep = keep[:, keep[0] != 4]            # node 4 has no outgoing edge -> empty segment
    sc["edge_indices"] = np.ascontiguousarray(keep)
    gen = {"edge_indices": sc["edge_indices"]}
    for aggr in ("max", "add", "mean"):
        ma = build_reference(1, aggr)
        load_formula_weights(ma, VLSATConfig(N_LAYERS=1, GCN_AGGR=aggr))
        ra = run(ma, synth.collate([sc]))
        for k in ("obj3d", "obj2d", "rel3d", "rel2d"):
            gen[f"{aggr}.{k}"] = ra[k]
    np.savez(os.path.join(HERE, "general_edges_n6_p32_l1.npz"), **gen)

    # ---- cfg 2 scene shape: 40 objects x 256 pts, L=3, one scene (seed 1000) and the
    #      first scene of the bench batch; outputs only (~375 KB)
    c2 = VLSATConfig(N_LAYERS=3)
    m3 = build_reference(3)
    load_formula_weights(m3, c2)
    r2 = run(m3, synth.make_batch(1, 40, 256, seed0=1000))
    np.savez_compressed(os.path.join(HERE, "cfg2_n40_p256_l3.npz"), **r2)

    # ---- P=1024 object encoder (cfg 5 point count) on 3 objects: encoder output only
    b5 = synth.make_batch(1, 3, 1024, seed0=5000)
    with torch.no_grad():
        f = m3.obj_encoder(torch.from_numpy(b5["obj_points"]))
    np.savez(os.path.join(HERE, "pointnet_n3_p1024.npz"), obj_encoder=f.numpy())

    for k, v in meta.items():
        print(k, v)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
