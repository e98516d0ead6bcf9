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


# ---- round-2 goldens (tests/golden/make_golden_r2.py) ---------------------------------------------------------------
RAGGED = lambda: _t(synth.collate([synth.make_scene(5, 64, 2000), synth.make_scene(7, 64, 2001)]))   # noqa: E731


def trained_adapter_weights(z, cfg):
    w = synth.make_weights(cfg)
    for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        w["clip_adapter." + k] = z["w.clip_adapter." + k]
    return w


def test_adapter_with_the_trained_checkpoint(golden_dir):
    """G7: the reference's only trained weights, clip_adapter/checkpoint/origin_mean.pth, through
    AdapterModel.forward alone and through the whole forward."""
    z = np.load(os.path.join(golden_dir, "adapter_trained.npz"))
    cfg = VLSATConfig(N_LAYERS=2)
    w = O.to_torch(trained_adapter_weights(z, cfg))
    _close(O.adapter(torch.from_numpy(z["adapter_x"]), w), z["adapter_y"], 2e-6, "adapter alone")
    b = _t(synth.make_batch(1, 8, 256, seed0=1000))
    taps = {}
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"], taps=taps)
    _close(taps["clip_adapter"], z["adapter_tap"], 2e-6, "adapter tap")
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[name], name=name)


@pytest.mark.parametrize("tag,scale", [("stress_x4", 4.0), ("stress_x025", 0.25)])
def test_trained_scale_stress(golden_dir, tag, scale):
    """At x4 the network amplifies fp32 roundoff: the reference's own outputs sit 4e-5..2e-4 from an fp64 evaluation and
    this fp32 restatement 4e-4 (different BLAS blocking), so the fp32 comparison uses the contract's 1e-3 and the fp64
    oracle is held to 3e-4 of the reference."""
    z = np.load(os.path.join(golden_dir, tag + ".npz"))
    cfg = VLSATConfig(N_LAYERS=3)
    w = O.to_torch(synth.make_weights_stress(cfg, scale))
    b = RAGGED()
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[name], 1e-3 if scale > 1 else 1e-5, f"{tag}.{name}")
    w64 = O.to_torch(synth.make_weights_stress(cfg, scale), torch.float64)
    out64 = O.forward(w64, cfg, b["obj_points"].double(), b["obj_2d_feats"].double(), b["edge_indices"], b["descriptor"].double(),
                      b["batch_ids"])
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out64):
        _close(o.float(), z[name], 3e-4 if scale > 1 else 1e-5, f"{tag}.{name} (fp64 oracle)")


def test_two_full_80_object_scenes_of_the_reference(golden_dir):
    """E = 6320 per scene through the reference itself: licenses the oracle (q-chunked attention) at a size with
    50 flash-attention query tiles."""
    z = np.load(os.path.join(golden_dir, "n80_p128_l3.npz"))
    cfg = VLSATConfig(N_LAYERS=3)
    w = O.to_torch(synth.make_weights(cfg))
    b = _t(synth.collate([synth.make_scene(80, 128, 8000), synth.make_scene(80, 128, 8001)]))
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    idx = torch.from_numpy(z["edge_idx"])
    _close(out[0], z["obj3d"], 5e-5, "obj3d")
    _close(out[1], z["obj2d"], 5e-5, "obj2d")
    _close(out[2][idx], z["rel3d"], 5e-5, "rel3d")
    _close(out[3][idx], z["rel2d"], 5e-5, "rel2d")


def test_train_outputs(golden_dir):
    """Mmgnet.forward(istrain=True), modules in eval mode: the four extras of the 8-tuple."""
    z = np.load(os.path.join(golden_dir, "train_outputs.npz"))
    cfg = VLSATConfig(N_LAYERS=2, train_outputs=True)
    w = O.to_torch(synth.make_weights(cfg))
    b = RAGGED()
    out = O.forward_train_outputs(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    names = ("obj3d", "obj2d", "rel3d", "rel2d", "obj_feature_3d_mimic", "obj_features_2d_mimic", "gcn_edge_feature_2d_dis")
    for name, o in zip(names, out[:7]):
        _close(o, z[name], name=name)
    assert abs(out[7] - float(z["logit_scale"])) < 1e-5


@pytest.mark.parametrize("h,a", [(4, 256), (16, 256), (8, 128), (8, 512)])
def test_num_heads_and_dim_atten(golden_dir, h, a):
    z = np.load(os.path.join(golden_dir, f"heads_h{h}_a{a}.npz"))
    cfg = VLSATConfig(N_LAYERS=2, NUM_HEADS=h, DIM_ATTEN=a)
    w = O.to_torch(synth.make_weights(cfg))
    b = RAGGED()
    out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
    for name, o in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out):
        _close(o, z[name], name=f"h{h}a{a}.{name}")
