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


def test_device_preparation_equals_the_reference_on_its_recorded_draws(golden_dir):
    """The device half (vlsat_prepare_objects) against the REAL data_preparation (reference dataset_3dssg.py:285-293), fixture
    tests/golden/scan_small.npz (made by make_golden_scan.py from the reference's dataset class): the same mesh file read by
    scan.read_ply, the draws np.random.choice made inside the reference, -> obj_points and descriptor within 1e-6 (the reference
    computes the descriptor in float64 and centres float32 points with their float32 mean; so does the kernel, up to the rounding
    of that mean)."""
    import os
    z = np.load(os.path.join(golden_dir, "scan_small.npz"))
    e = json.load(open(os.path.join(golden_dir, "scan_small_expect.json")))
    mesh = S.read_ply(os.path.join(golden_dir, "scan_small.ply"))
    rel, objs, scans = S.read_relationships(os.path.join(golden_dir, "scan_small_relationships.json"), ["scan-a"])
    rel, objs = rel["scan-a_0"], objs["scan-a_0"]
    nodes = S.scene_nodes(mesh["instances"], objs)
    assert nodes == e["nodes_scan_a_0"]
    from vlsat_amd import prep
    for tag, use_extra in (("prep_multi_all_xyz", False), ("prep_multi_all_xyz_rgb_normal", True)):
        choice = np.stack([np.where(mesh["instances"] == i)[0][z[tag + "_choice"][k]] for k, i in enumerate(nodes)]).astype(np.int32)
        d_xyz = torch.from_numpy(np.ascontiguousarray(mesh["points"], dtype=np.float32)).to(DEV)
        d_choice = torch.from_numpy(choice).to(DEV)
        obj, desc = prep.prepare_objects(d_xyz, d_choice)
        want = torch.from_numpy(z[tag + "_obj_points"]).permute(0, 2, 1)                 # reference layout [N,P,C] -> [N,C,P]
        assert float((obj.cpu() - want[:, :3]).abs().max()) <= 1e-6
        assert float((desc.cpu() - torch.from_numpy(z[tag + "_descriptor"])).abs().max()) <= 1e-6
    # the whole entry on the same files: everything but the (library-drawn) selection equals the reference's outputs
    b = S.prepare_scan(os.path.join(golden_dir, "scan_small.ply"), objs, e["classes"], rel, e["relations"], num_points=16, seed=1, device=DEV,
                       use_rgb=True, use_normal=True, feature_loader=lambda i, name: z["multi_view_feats"][nodes.index(i)])
    assert b["instance_ids"] == nodes and b["obj_points"].shape == (5, 9, 16)
    assert np.array_equal(b["edge_indices"].cpu().numpy(), z["prep_multi_all_xyz_edge_indices"])
    assert np.array_equal(b["gt_class"].cpu().numpy(), z["prep_multi_all_xyz_label_node"])
    assert np.array_equal(b["gt_rel_cls"].cpu().numpy(), z["prep_multi_all_xyz_gt_rels"])
    assert np.array_equal(b["obj_2d_feats"].cpu().numpy(), z["prep_multi_all_xyz_obj_2d_feats"])
    assert b["points_per_instance"].cpu().tolist() == z["prep_multi_all_xyz_count"].tolist()
    ch = b["choice"].cpu().numpy()
    extra = z["mesh_points_xyz_rgb_normal"][:, 3:][ch].astype(np.float32).transpose(0, 2, 1)
    assert np.array_equal(b["obj_points"][:, 3:].cpu().numpy(), extra)                     # colour / normal channels: gathered, not centred
    big = {**mesh, "instances": np.where(mesh["instances"] == 5, 1 << 24, mesh["instances"])}      # an instance id the id map cannot hold
    with pytest.raises(S.ScanError, match="distinct"):
        S.prepare_scan(big, {**objs, (1 << 24): "chair"}, e["classes"], rel, e["relations"], 16, 1, device=DEV)
