"""Real-data entry on the device (vlsat_amd/scan.py::prepare_scan): a label mesh + 3DSSG annotations -> the batch the forward and the
evaluation loop take.  Needs an MI355X."""
import json

import numpy as np
import pytest
import torch

import vlsat_amd  # noqa: F401
from vlsat_amd import VLSATConfig, scan as S, synth
from test_scan_cpu import make_scene, relationships_doc, write_ply

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scan(tmp_path, binary=True):
    pts, rgb, inst = make_scene(3, 4000, ids=(1, 2, 5, 9, 12))
    ply = str(tmp_path / "labels.instances.align.annotated.v2.ply")
    write_ply(ply, pts, rgb, inst, binary)
    jp = str(tmp_path / "relationships_validation.json")
    json.dump(relationships_doc(), open(jp, "w"))
    rel, objs, scans = S.read_relationships(jp, ["s-a"])
    return ply, pts, rgb, inst, rel[scans[0]], objs[scans[0]]


CLASSES = ["bed", "chair", "floor", "lamp", "sofa", "table", "wall"] + ["c%d" % i for i in range(153)]       # 160 names
RELS = ["attached to", "close by", "left", "standing on"] + ["r%d" % i for i in range(22)]                   # 26 names


def test_prepare_scan_equals_the_oracle_preparation_of_the_same_selection(tmp_path):
    from oracle import prep_oracle as PO
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    ply, pts, rgb, inst, rel, objs = _scan(tmp_path)
    b = S.prepare_scan(ply, objs, CLASSES, rel, RELS, num_points=128, seed=11, device=DEV)
    assert b["instance_ids"] == [5, 1, 9, 2] and b["fc_sizes"] == [4]
    choice = b["choice"].cpu().numpy()
    for k, i in enumerate(b["instance_ids"]):
        assert np.all(inst[choice[k]] == i)                                   # every drawn vertex belongs to the object
        assert int(b["points_per_instance"][k]) == int((inst == i).sum())
    ref_pts, ref_desc = PO.prepare_objects(pts.astype(np.float32), choice)
    assert float((b["obj_points"].cpu() - ref_pts).abs().max()) < 1e-5
    assert float((b["descriptor"].cpu() - ref_desc).abs().max()) < 2e-4
    nodes, edges, gt_class, gt_rel = PO.scene_labels(inst, objs, CLASSES, rel, RELS, True, True)
    assert np.array_equal(b["edge_indices"].cpu().numpy(), edges) and np.array_equal(b["gt_class"].cpu().numpy(), gt_class)
    assert np.array_equal(b["gt_rel_cls"].cpu().numpy(), gt_rel)
    again = S.prepare_scan(ply, objs, CLASSES, rel, RELS, num_points=128, seed=11, device=DEV)
    assert torch.equal(again["obj_points"], b["obj_points"])                   # same seed, same selection
    other = S.prepare_scan(ply, objs, CLASSES, rel, RELS, num_points=128, seed=12, device=DEV)
    assert not torch.equal(other["choice"], b["choice"])


def test_a_prepared_scan_runs_through_forward_and_the_evaluation_loop(tmp_path):
    from vlsat_amd import evaluate as EV
    from vlsat_amd.model import VLSATModel
    ply, pts, rgb, inst, rel, objs = _scan(tmp_path, binary=False)
    cfg = VLSATConfig(N_LAYERS=2)
    m = VLSATModel(cfg, DEV).load_state(synth.make_weights(cfg)).eval()
    feats = np.random.default_rng(0).normal(size=(100, 512)).astype(np.float32)
    b = S.prepare_scan(ply, objs, CLASSES, rel, RELS, num_points=64, seed=3, device=DEV, feature_loader=lambda i, name: feats[i])
    assert torch.equal(b["obj_2d_feats"].cpu(), torch.from_numpy(feats[[5, 1, 9, 2]]))
    out = m(b["obj_points"], b["obj_2d_feats"], b["edge_indices"].t().contiguous(), b["descriptor"], b["batch_ids"])
    assert out[0].shape == (4, 160) and out[2].shape == (12, 26) and all(bool(torch.isfinite(o).all()) for o in out)
    res0 = EV.validation(m, [b, b])                              # reference-compatible loop
    res1 = EV.validation(m, [b, b], device=DEV, workers=1)       # counts on the device
    assert res0.keys() == res1.keys() and all(abs(res0[k] - res1[k]) < 1e-9 for k in res0), (res0, res1)
    # annotated pairs only (all_edge = False): five of the six annotations connect objects that own points
    sparse = S.prepare_scan(ply, objs, CLASSES, rel, RELS, num_points=64, seed=3, device=DEV, all_edge=False)
    assert sparse["edge_indices"].shape == (5, 2) and "fc_sizes" not in sparse
    out = m(sparse["obj_points"], sparse["obj_2d_feats"], sparse["edge_indices"].t().contiguous(), sparse["descriptor"], sparse["batch_ids"])
    assert out[2].shape == (5, 26)


def test_colour_channels_ride_along_with_the_selection(tmp_path):
    ply, pts, rgb, inst, rel, objs = _scan(tmp_path)
    b = S.prepare_scan(ply, objs, CLASSES, rel, RELS, num_points=32, seed=5, device=DEV, use_rgb=True)
    assert b["obj_points"].shape == (4, 6, 32)
    choice = b["choice"].cpu().numpy()
    want = (rgb[choice] / 255.0).astype(np.float32).transpose(0, 2, 1)
    assert np.allclose(b["obj_points"][:, 3:].cpu().numpy(), want, atol=1e-7)
