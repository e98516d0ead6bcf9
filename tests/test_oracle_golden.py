"""The CPU oracle (oracle/vlsat_oracle.py) against golden vectors produced by the REAL
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import vlsat_amd  # noqa: F401
from vlsat_amd import VLSATConfig, synth
from oracle import vlsat_oracle as O

TOL = 2e-5   # fp32 oracle vs fp32 reference: same op order up to BLAS blocking


def _t(b):
    return {k: torch.from_numpy(v) for k, v in b.items()}


def _close(a, b, tol=TOL, name=""):
    a = a.numpy() if torch.is_tensor(a) else a
    err = float(np.abs(a - b).max())
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert err <= tol, f"{name}: max-abs-err {err:.3e} > {tol}"


def test_kat_gather_scatter(golden_dir):
    """reference network_util.py:75-99 evaluated through the reference classes."""
    z = np.load(os.path.join(golden_dir, "kat_index.npz"))
    x = torch.zeros(3, 5)
    x[1], x[2] = 1, 2
    ei = torch.tensor([[0, 1, 2], [2, 1, 0]])
    tmp = -torch.arange(5, dtype=torch.float32)[:, None].repeat(1, 2)
    ei2 = torch.tensor([[0, 1, 2, 1, 0], [2, 1, 1, 1, 1]])
    for flow in ("source_to_target", "target_to_source"):
        xi, xj = O.gen_index(x, ei, flow)
        assert np.array_equal(xi.numpy(), z[f"gen_index.{flow}.x_i"])
        assert np.array_equal(xj.numpy(), z[f"gen_index.{flow}.x_j"])
        for aggr in ("max", "add", "mean"):
            got = O.aggre_index(tmp, ei2, 3, aggr, flow).numpy()
            assert np.array_equal(got, z[f"aggre_index.{flow}.{aggr}"]), (flow, aggr)
    # the values SURVEY §4 derives by hand from PyG semantics
    assert O.aggre_index(tmp, ei2, 3, "max", "target_to_source")[:, 0].tolist() == [0.0, -1.0, -2.0]
    assert O.aggre_index(tmp, ei2, 3, "add", "target_to_source")[:, 0].tolist() == [-4.0, -4.0, -2.0]


def test_cfg1_full_and_taps(golden_dir):
    z = np.load(os.path.join(golden_dir, "cfg1_n8_p256_l2.npz"))
    cfg = VLSATConfig(N_LAYERS=2)
    w = O.to_torch(synth.make_weights(cfg))
    b = _t(synth.make_batch(1, 8, 256, seed0=1000))
    taps = {}
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"],
                    b["batch_ids"], taps=taps)
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[name], name=name)
    _close(taps["obj_encoder"], z["tap.obj_encoder"], name="obj_encoder")
    _close(taps["node_embed"][:, :504], z["tap.mlp_3d"], name="mlp_3d")
    _close(taps["edge_descriptor"], z["tap.edge_descriptor"], 1e-6, "edge_descriptor")
    _close(taps["rel_encoder_3d"], z["tap.rel_encoder_3d"], name="rel_encoder_3d")
    _close(taps["rel_encoder_2d"], z["tap.rel_encoder_2d"], name="rel_encoder_2d")
    _close(taps["clip_adapter"], z["tap.clip_adapter"], name="adapter")
    _close(taps["dist_bias"].permute(1, 2, 0)[None], z["tap.dist_bias"], name="dist_bias")
    _close(taps["self_attn0"][None], z["tap.self_attn0"], name="self_attn0")
    _close(taps["cross_attn0"][None], z["tap.cross_attn0"], name="cross_attn0")
    _close(taps["gcn3d0.gated"], z["tap.edgeatten3d0.0"], name="gated")
    _close(taps["gcn3d0.prob"], z["tap.edgeatten3d0.2"], name="prob (head layout)")
    _close(taps["gcn3d0.edge"], z["tap.edgeatten3d0.1"], name="edge'")
    _close(taps["gcn3d0.node"], z["tap.gcn3d0.0"], name="gcn3d node")
    _close(taps["gcn2d0.node"], z["tap.gcn2d0.0"], name="gcn2d node")
    _close(taps["gcn2d0.edge"], z["tap.gcn2d0.1"], name="gcn2d edge")
    _close(taps["cross_attn_rel0"][None], z["tap.cross_attn_rel0"], name="cross_attn_rel0")
    for i in range(4):
        _close(taps[f"mmg.{i}"], z[f"tap.mmg.{i}"], name=f"mmg.{i}")


def test_ragged_batch_per_scene_contract(golden_dir):
    z = np.load(os.path.join(golden_dir, "ragged_n5_n7_p64_l2.npz"))
    cfg = VLSATConfig(N_LAYERS=2)
    w = O.to_torch(synth.make_weights(cfg))
    b = _t(synth.collate([synth.make_scene(5, 64, 2000), synth.make_scene(7, 64, 2001)]))
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[name], name=name)
    # the reference's own batched call agrees on the (batch-independent) 3D branch
    _close(out[0], z["batched_obj3d"], name="batched obj3d")
    _close(out[2], z["batched_rel3d"], name="batched rel3d")


@pytest.mark.parametrize("aggr", ["max", "add", "mean"])
def test_general_edges_all_aggregators(golden_dir, aggr):
    z = np.load(os.path.join(golden_dir, "general_edges_n6_p32_l1.npz"))
    cfg = VLSATConfig(N_LAYERS=1, GCN_AGGR=aggr)
    w = O.to_torch(synth.make_weights(cfg))
    sc = synth.make_scene(6, 32, 3000)
    sc["edge_indices"] = z["edge_indices"]
    b = _t(synth.collate([sc]))
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[f"{aggr}.{name}"], name=f"{aggr}.{name}")


def test_cross_scene_batch_call_of_the_reference(golden_dir):
    """One reference call carrying two scenes: the edge cross-attention spans the whole batch, so the 2D
    outputs differ from the per-scene ones; the 3D outputs do not."""
    z = np.load(os.path.join(golden_dir, "ragged_n5_n7_p64_l2.npz"))
    cfg = VLSATConfig(N_LAYERS=2)
    w = O.to_torch(synth.make_weights(cfg))
    b = _t(synth.collate([synth.make_scene(5, 64, 2000), synth.make_scene(7, 64, 2001)]))
    out = O.forward_cross_scene(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    for name, o in zip(("batched_obj3d", "batched_obj2d", "batched_rel3d", "batched_rel2d"), out):
        _close(o, z[name], name=name)
    assert float(np.abs(z["batched_rel2d"] - z["rel2d"]).max()) > 1e-3          # the two contracts really differ
    _close(out[0], z["obj3d"], name="3D branch is batch-independent")


@pytest.mark.parametrize("case", sorted(synth.SWITCH_CASES))
def test_config_switches(golden_dir, case):
    """WITH_BN / USE_GCN_EDGE=false / multi_rel_outputs=false / USE_RGB+USE_NORMAL (SURVEY 8a switch table) against
    the real reference run with that switch (tests/golden/make_golden_switches.py)."""
    z = np.load(os.path.join(golden_dir, case + ".npz"))
    cfg = VLSATConfig(**synth.SWITCH_CASES[case])
    w = O.to_torch(synth.make_weights(cfg))
    b = _t(synth.collate(synth.switch_scenes(cfg)))
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[name], name=f"{case}.{name}")


def test_cfg2_scene_shape(golden_dir):
    z = np.load(os.path.join(golden_dir, "cfg2_n40_p256_l3.npz"))
    cfg = VLSATConfig(N_LAYERS=3)
    w = O.to_torch(synth.make_weights(cfg))
    b = _t(synth.make_batch(1, 40, 256, seed0=1000))
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[name], 5e-5, name)


def test_pointnet_p1024(golden_dir):
    z = np.load(os.path.join(golden_dir, "pointnet_n3_p1024.npz"))
    cfg = VLSATConfig(N_LAYERS=3)
    w = O.to_torch(synth.make_weights(cfg))
    b = _t(synth.make_batch(1, 3, 1024, seed0=5000))
    _close(O.pointnet_feat(b["obj_points"], w, "obj_encoder"), z["obj_encoder"], name="pointnet P=1024")


def test_chunked_attention_equals_unchunked():
    """Licenses the q-chunked attention as the cfg-5 oracle."""
    cfg = VLSATConfig(N_LAYERS=1)
    w = O.to_torch(synth.make_weights(cfg))
    g = torch.Generator().manual_seed(3)
    e2, e3 = torch.randn(300, 512, generator=g), torch.randn(300, 512, generator=g)
    a = O.mha(e2, e3, w, "mmg.cross_attn_rel.0", 8, q_chunk=4096)
    b = O.mha(e2, e3, w, "mmg.cross_attn_rel.0", 8, q_chunk=64)
    assert float((a - b).abs().max()) < 1e-5
