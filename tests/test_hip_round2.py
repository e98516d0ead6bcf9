"""Round-2 parity cases of the HIP path (through VLSATModel -> C ABI): trained adapter checkpoint, trained-scale
stress weights, two full 80-object scenes of the reference, forward(istrain=True), the content-keyed plan cache,
weight reloading.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

import vlsat_amd  # noqa: F401
from vlsat_amd import VLSATConfig, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ("obj3d", "obj2d", "rel3d", "rel2d")
TOL, TIGHT = 1e-3, 1e-4


def _dev(b):
    return {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}


def _model(cfg, weights):
    from vlsat_amd.model import VLSATModel
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    return VLSATModel(cfg, DEV).load_state(weights).eval()


def _run(m, b, **kw):
    d = _dev(b)
    out = m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"], **kw)
    torch.cuda.synchronize()
    return [o.cpu() for o in out]


def _check(got, ref, tol, what, names=NAMES):
    errs = {}
    for n, g, r in zip(names, got, ref):
        r = torch.as_tensor(r).float()
        assert g.shape == r.shape, (what, n, g.shape, r.shape)
        assert torch.isfinite(g).all(), f"{what} {n}: non-finite output"
        errs[n] = float((g - r).abs().max()) if g.numel() else 0.0
    print(what, {k: f"{v:.2e}" for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, f"{what}: max-abs-err over {tol}: {bad}"
    return errs


RAGGED = lambda: synth.collate([synth.make_scene(5, 64, 2000), synth.make_scene(7, 64, 2001)])   # noqa: E731


def test_trained_adapter_checkpoint(golden_dir):
    """G7: the reference's trained clip_adapter weights (origin_mean.pth) on the halved-W2 / residual-as-accumulator
    algebra of the HIP adapter, against the reference's own run."""
    z = np.load(os.path.join(golden_dir, "adapter_trained.npz"))
    cfg = VLSATConfig(N_LAYERS=2)
    w = synth.make_weights(cfg)
    for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        w["clip_adapter." + k] = z["w.clip_adapter." + k]
    m = _model(cfg, w)
    b = synth.make_batch(1, 8, 256, seed0=1000)
    _check(_run(m, b), [z[n] for n in NAMES], TIGHT, "trained adapter, full forward vs reference")
    d = _dev(b)
    m.debug_stop_after(4)
    try:
        m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
        x2 = m.debug_buffer(d["edge_indices"], d["batch_ids"], 8, 256, "X2").cpu()
    finally:
        m.debug_stop_after(-1)
    err = float((x2 - torch.from_numpy(z["adapter_tap"])).abs().max())
    assert err < 2e-6, f"adapter output vs reference: {err:.3e}"
    m.close()


@pytest.mark.parametrize("tag,scale", [("stress_x4", 4.0), ("stress_x025", 0.25)])
def test_trained_scale_stress(golden_dir, tag, scale):
    """GCN matrices x4 / x0.25, LayerNorm gains in U(0.3, 3): the hoisted nn_edge.0 / proj_query / gate-layer-1
    algebra away from Xavier scale.  At x4 the network amplifies fp32 roundoff (the reference itself sits up to 2e-4
    from an fp64 evaluation, tests/test_oracle_golden.py), so the contract's 1e-3 is checked against the reference
    golden AND against the fp64 oracle."""
    from oracle import vlsat_oracle as O
    z = np.load(os.path.join(golden_dir, tag + ".npz"))
    cfg = VLSATConfig(N_LAYERS=3)
    wn = synth.make_weights_stress(cfg, scale)
    m = _model(cfg, wn)
    b = RAGGED()
    got = _run(m, b)
    tol = TOL if scale > 1 else 1e-5
    _check(got, [z[n] for n in NAMES], tol, f"{tag} vs reference golden")
    c = {k: torch.from_numpy(v) for k, v in b.items()}
    ref64 = O.forward(O.to_torch(wn, torch.float64), cfg, c["obj_points"].double(), c["obj_2d_feats"].double(),
                      c["edge_indices"], c["descriptor"].double(), c["batch_ids"])
    _check(got, ref64, tol, f"{tag} vs fp64 oracle")
    m.close()


def test_two_full_80_object_scenes_vs_reference(golden_dir):
    """E = 6320 per scene (50 flash-attention query tiles, several GEMM rounds), L=3, pinned to the reference itself."""
    z = np.load(os.path.join(golden_dir, "n80_p128_l3.npz"))
    cfg = VLSATConfig(N_LAYERS=3)
    m = _model(cfg, synth.make_weights(cfg))
    got = _run(m, synth.collate([synth.make_scene(80, 128, 8000), synth.make_scene(80, 128, 8001)]))
    idx = torch.from_numpy(z["edge_idx"])
    _check([got[0], got[1], got[2][idx], got[3][idx]], [z[n] for n in NAMES], TIGHT, "2 x 80 objects vs reference")
    m.close()


@pytest.mark.parametrize("pv_terms", [3, 2])
def test_bf16x3_on_stress_weights_and_large_scenes(golden_dir, pv_terms):
    """The split-bf16 mode where its shortcuts could show: weights x4 with LayerNorm gains up to 3 (peaked attention, large
    logits) and two 80-object scenes (6320-token attention), with the P.V product of the edge attention at 3 terms and at 2
    (probabilities single-rounded, V exact: debug option flash_pv_terms).  The x4 network amplifies roundoff by ~4000 (fp32
    itself ends 4.4e-4 from fp64 there, tests/test_oracle_golden.py): the 16-bit significands of the split operands give
    ~2e-3, inside the 1e-2 of BASELINE configs[2] but not inside the fp32 config's 1e-3, which this mode only meets at the
    weight scale of the bench (4e-5).  Unit-scale weights, 6320-token scenes: 2e-4."""
    from oracle import vlsat_oracle as O
    cfg = VLSATConfig(N_LAYERS=3)
    wn = synth.make_weights_stress(cfg, 4.0)
    m = _model(cfg, wn).set_gemm_precision("bf16x3").debug_option("flash_pv_terms", pv_terms)
    b = RAGGED()
    c = {k: torch.from_numpy(v) for k, v in b.items()}
    ref64 = O.forward(O.to_torch(wn, torch.float64), cfg, c["obj_points"].double(), c["obj_2d_feats"].double(),
                      c["edge_indices"], c["descriptor"].double(), c["batch_ids"])
    _check(_run(m, b), ref64, 5e-3, f"bf16x3 (P.V {pv_terms} terms), stress x4 vs fp64 oracle")
    m.close()
    z = np.load(os.path.join(golden_dir, "n80_p128_l3.npz"))
    m = _model(cfg, synth.make_weights(cfg)).set_gemm_precision("bf16x3").debug_option("flash_pv_terms", pv_terms)
    got = _run(m, synth.collate([synth.make_scene(80, 128, 8000), synth.make_scene(80, 128, 8001)]))
    idx = torch.from_numpy(z["edge_idx"])
    _check([got[0], got[1], got[2][idx], got[3][idx]], [z[n] for n in NAMES], 2e-4, f"bf16x3 (P.V {pv_terms} terms), 2 x 80 objects vs reference")
    m.close()


def test_train_outputs_golden(golden_dir):
    """forward(istrain=True): the reference's 8-tuple (eval-mode modules), ragged 2-scene batch."""
    z = np.load(os.path.join(golden_dir, "train_outputs.npz"))
    cfg = VLSATConfig(N_LAYERS=2, train_outputs=True)
    m = _model(cfg, synth.make_weights(cfg))
    d = _dev(RAGGED())
    out = m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"], istrain=True)
    assert len(out) == 8
    names = NAMES + ("obj_feature_3d_mimic", "obj_features_2d_mimic", "gcn_edge_feature_2d_dis")
    _check([o.cpu() for o in out[:7]], [z[n] for n in names], TIGHT, "istrain=True", names)
    assert abs(float(out[7]) - float(z["logit_scale"])) < 1e-4
    # the same call without the flag still returns the 4-tuple, identical in its first four entries
    four = m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
    assert len(four) == 4 and all(torch.equal(a, b) for a, b in zip(four, out[:4]))
    m.close()


def test_plan_cache_is_keyed_by_graph_content():
    """A fresh tensor with the same graph (what process_val's .t().contiguous() produces every call) re-uses the plan;
    host-resident edge lists need no device copy; fc_sizes needs neither; different graphs never share a plan."""
    cfg = VLSATConfig(N_LAYERS=2)
    m = _model(cfg, synth.make_weights(cfg))
    b = synth.make_batch(1, 9, 64, seed0=31)
    d = _dev(b)
    args = lambda ei, bid: (d["obj_points"], d["obj_2d_feats"], ei, d["descriptor"], bid)   # noqa: E731
    ref = [o.clone() for o in m(*args(d["edge_indices"], d["batch_ids"]))]
    s0 = dict(m.plan_stats)
    assert s0["builds"] == 1
    for _ in range(3):                                               # same tensor objects: identity hit
        m(*args(d["edge_indices"], d["batch_ids"]))
    assert m.plan_stats["identity_hits"] == s0["identity_hits"] + 3 and m.plan_stats["builds"] == 1
    for _ in range(3):                                               # fresh device tensors, same content
        out = m(*args(d["edge_indices"].clone(), d["batch_ids"].clone() + 5))
        assert all(torch.equal(a, c) for a, c in zip(out, ref))
    assert m.plan_stats["builds"] == 1 and m.plan_stats["hits"] >= 3
    copies = m.plan_stats["d2h_copies"]
    out = m(*args(torch.from_numpy(b["edge_indices"]), torch.from_numpy(b["batch_ids"])))     # host tensors
    assert all(torch.equal(a, c) for a, c in zip(out, ref)) and m.plan_stats["d2h_copies"] == copies
    out = m(*args(d["edge_indices"].clone(), None), fc_sizes=[9])                           # declared FC graph
    assert all(torch.equal(a, c) for a, c in zip(out, ref)) and m.plan_stats["d2h_copies"] == copies
    assert m.plan_stats["builds"] == 2                               # ("fc", sizes) is its own key, built once
    m(*args(d["edge_indices"].clone(), None), fc_sizes=[9])
    assert m.plan_stats["builds"] == 2
    # a different graph of the same shape must not hit
    other = d["edge_indices"].clone()
    other[:, [0, 1]] = other[:, [1, 0]]
    out2 = m(*args(other, d["batch_ids"]))
    assert m.plan_stats["builds"] == 3
    assert float((out2[2][0] - ref[2][1]).abs().max()) < 1e-5 and float((out2[2][1] - ref[2][0]).abs().max()) < 1e-5
    with pytest.raises(Exception):
        m(*args(d["edge_indices"], None), fc_sizes=[8])
    m.close()


def test_many_scene_sizes_in_a_row_recycle_workspaces():
    """60 scenes of varying size, one per call, more distinct graphs than the cache holds: plans are evicted and their
    arenas recycled behind events (no device-wide sync); every result equals the oracle's."""
    from oracle import vlsat_oracle as O
    cfg = VLSATConfig(N_LAYERS=2)
    wn = synth.make_weights(cfg)
    m = _model(cfg, wn)
    m.MAX_PLANS = 5
    w = O.to_torch(wn)
    g = np.random.default_rng(3)
    outs, refs = [], []
    for i in range(60):
        n = int(g.integers(2, 24))
        b = synth.collate([synth.make_scene(n, 32, 15000 + i)])
        d = _dev(b)
        outs.append(m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"]))
        if i % 6 == 0:
            c = {k: torch.from_numpy(v) for k, v in b.items()}
            refs.append((i, O.forward(w, cfg, c["obj_points"], c["obj_2d_feats"], c["edge_indices"], c["descriptor"], c["batch_ids"])))
    torch.cuda.synchronize()
    for i, ref in refs:
        _check([o.cpu() for o in outs[i]], ref, TIGHT, f"scene {i}")
    assert len(m._plans) <= 5
    m.close()


def test_weights_can_be_reloaded():
    """BaseModel.load may be called repeatedly (model_base.py:75-129); so may load_state, also in a bf16 mode."""
    from oracle import vlsat_oracle as O
    cfg = VLSATConfig(N_LAYERS=2)
    w0, w1 = synth.make_weights(cfg, 0), synth.make_weights(cfg, 1)
    m = _model(cfg, w0)
    b = RAGGED()
    c = {k: torch.from_numpy(v) for k, v in b.items()}
    ref = lambda w: O.forward(O.to_torch(w), cfg, c["obj_points"], c["obj_2d_feats"], c["edge_indices"], c["descriptor"], c["batch_ids"])   # noqa: E731
    _check(_run(m, b), ref(w0), TIGHT, "first weights")
    m.load_state(w1)
    _check(_run(m, b), ref(w1), TIGHT, "reloaded weights")
    m.set_gemm_precision("bf16x3")
    m.load_state(w0)
    _check(_run(m, b), ref(w0), TOL, "reloaded in bf16x3 mode")
    m.close()


@pytest.mark.parametrize("h,a", [(4, 256), (16, 256), (8, 128), (8, 512)])
@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16_mixed"])
def test_num_heads_and_dim_atten_golden(golden_dir, h, a, mode):
    """MODEL.NUM_HEADS / MODEL.DIM_ATTEN other than the shipped 8 / 256 (reference network_MMG.py:48-50): head dims 128 / 32,
    gate output widths 64 / 16, through the generic per-head kernels, against the real reference run with that config."""
    z = np.load(os.path.join(golden_dir, f"heads_h{h}_a{a}.npz"))
    cfg = VLSATConfig(N_LAYERS=2, NUM_HEADS=h, DIM_ATTEN=a)
    m = _model(cfg, synth.make_weights(cfg)).set_gemm_precision(mode)
    tol = {"fp32": TIGHT, "bf16x3": TOL, "bf16_mixed": 1e-2}[mode]          # (BASELINE configs[2] tolerance for the single-rounding mode)
    _check(_run(m, RAGGED()), [z[n] for n in NAMES], tol, f"H={h} A={a} {mode} vs reference golden")
    m.close()


def test_native_rccl_allreduce_single_rank():
    """vlsat_comm_* / vlsat_metrics_allreduce (include/vlsat.h): the library's own RCCL entry point for the one
    collective of the path.  One GPU here, so a communicator of one rank: the call chain (run-time RCCL resolution, unique
    id, init, all-reduce on the caller's stream, destroy) runs for real and the sum over one rank is the input itself.
    The same through bench.py's switch: checksums must equal the torch.distributed route."""
    import json
    import subprocess
    import sys
    from vlsat_amd import dist as vdist
    comm = vdist.NativeComm(0, 1)
    try:
        v = torch.arange(1, 488, dtype=torch.float64, device=DEV) * 0.5
        ref = v.clone()
        out = comm.allreduce(v)
        torch.cuda.synchronize()
        assert out.data_ptr() == v.data_ptr() and torch.equal(v, ref)
        with pytest.raises(Exception):
            comm.allreduce(torch.zeros(4, device=DEV))                       # fp32: rejected, not reinterpreted
    finally:
        comm.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ([], ["--native-allreduce"]):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--scenes", "4", "--steps", "1", "--warmup", "1", "--no-cpu",
                            "--no-profile", *extra], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    assert outs[1]["allreduce"].startswith("vlsat_metrics_allreduce")
    assert outs[0]["metrics_allreduced"] == outs[1]["metrics_allreduced"]


