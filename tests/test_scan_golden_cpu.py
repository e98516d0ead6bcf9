"""Host half of the input preparation (vlsat_amd/scan.py, oracle/prep_oracle.py) against what the REFERENCE's own dataset code
returns on the same files: tests/golden/scan_small.* are made by tests/golden/make_golden_scan.py, which imports
src/dataset/dataset_3dssg.py, src/dataset/DataLoader.py, utils/util.py and utils/util_ply.py from the reference.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import vlsat_amd  # noqa: F401
from vlsat_amd import scan as S, synth
from oracle import prep_oracle as PO

BAD = S._BAD_V2_SCAN


@pytest.fixture(scope="module")
def gold(golden_dir):
    z = np.load(os.path.join(golden_dir, "scan_small.npz"))
    e = json.load(open(os.path.join(golden_dir, "scan_small_expect.json")))
    return z, e, golden_dir


def _scan_a(gold):
    z, e, d = gold
    rel, objs, scans = S.read_relationships(os.path.join(d, "scan_small_relationships.json"), ["scan-a", BAD])
    return rel["scan-a_0"], objs["scan-a_0"]


@pytest.mark.parametrize("name", ["scan_small.ply", "scan_small_ascii.ply"])
def test_read_ply_equals_load_mesh(gold, name):
    """load_mesh (dataset_3dssg.py:38-58) through the trimesh stand-in vs read_ply + scene_points, binary and ASCII encodings of the
    same mesh: float64 vertices, rgb / 255, normals, instance ids -- bit for bit."""
    z, e, d = gold
    m = S.read_ply(os.path.join(d, name))
    want = z["mesh_points_xyz_rgb_normal"]
    assert m["points"].dtype == np.float64 and np.array_equal(m["points"], want[:, :3])
    assert np.array_equal(m["instances"], z["mesh_instances"]) and m["instances"].dtype == np.int64
    assert np.array_equal(S.scene_points(m, True, True), want)
    assert np.array_equal(S.scene_points(m, False, True), want[:, [0, 1, 2, 6, 7, 8]])
    assert np.array_equal(S.scene_points(m, True, False), want[:, :6])


def test_read_labels_fallback_to_label_property(gold, tmp_path):
    """util_ply.read_labels (:8-14): objectId, else label."""
    z, e, d = gold
    p = str(tmp_path / "l.ply")
    open(p, "wb").write(b"ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty ushort label\n"
                        b"end_header\n0 0 0 4\n0 0 0 0\n0 0 0 9\n0 0 0 9\n")
    assert S.read_ply(p)["instances"].tolist() == e["read_labels_fallback"]


def test_name_lists_read_like_the_reference(gold):
    """util.read_txt_to_list / read_relationships (utils/util.py:15-21,34-40): rstrip + lower, empty lines kept as entries."""
    z, e, d = gold
    got = S.read_name_list(os.path.join(d, "scan_names_messy.txt"))
    assert got == e["messy_read_txt_to_list"] == e["messy_read_relationships"]
    assert "" in got and "  lamp" in got and "floor" in got
    assert S.read_name_list(os.path.join(d, "3dssg_classes.txt")) == e["classes"]
    assert S.read_name_list(os.path.join(d, "3dssg_relations.txt")) == e["relations"]


@pytest.mark.parametrize("tag,label_file", [("v2", "labels.instances.align.annotated.v2.ply"), ("v1", "labels.instances.align.annotated.ply")])
def test_read_relationships_equals_read_relationship_json(gold, tag, label_file):
    """SSGDatasetGraph.read_relationship_json (:215-243): keys, scan order, object-map ORDER (the node order follows it), the
    scan the v2 label file skips."""
    z, e, d = gold
    want = e["relationships_" + tag]
    rel, objs, scans = S.read_relationships(os.path.join(d, "scan_small_relationships.json"), ["scan-a", BAD], label_file=label_file)
    assert scans == want["scans"]
    assert rel == want["rel"]
    assert {k: [[i, n] for i, n in v.items()] for k, v in objs.items()} == want["objs"]


@pytest.mark.parametrize("multi", [True, False])
@pytest.mark.parametrize("all_edge", [True, False])
def test_nodes_edges_and_labels_equal_data_preparation(gold, multi, all_edge):
    """data_preparation (:244-336) on the same mesh and annotations: node order = the object map's order restricted to instances
    that own vertices (NOT the sorted ids), the edge list (annotated pairs keep their duplicates), class labels, relation labels
    (multi-hot float32 [E,26] | int64 [E] where the later annotation of a pair wins)."""
    z, e, d = gold
    rel, objs = _scan_a(gold)
    tag = f"prep_{'multi' if multi else 'single'}_{'all' if all_edge else 'annot'}_xyz"
    names = e["relations"] if multi else ["none"] + e["relations"]
    inst = z["mesh_instances"]
    nodes = S.scene_nodes(inst, objs)
    assert nodes == e["nodes_scan_a_0"] and nodes != sorted(nodes)
    edges = S.edge_list(nodes, rel, all_edge)
    assert edges.dtype == np.int64 and np.array_equal(edges, z[tag + "_edge_indices"])
    gt_class, gt_rel = S.ground_truth(nodes, edges, objs, e["classes"], rel, names, multi)
    assert np.array_equal(gt_class, z[tag + "_label_node"]) and gt_class.dtype == z[tag + "_label_node"].dtype
    assert gt_rel.dtype == z[tag + "_gt_rels"].dtype and np.array_equal(gt_rel, z[tag + "_gt_rels"])
    # the oracle's loop-by-loop restatement, same fixtures
    o_nodes, o_edges, o_class, o_rel = PO.scene_labels(inst, objs, e["classes"], rel, names, multi, all_edge)
    assert o_nodes == nodes and np.array_equal(o_edges, edges) and np.array_equal(o_class, gt_class)
    assert o_rel.dtype == gt_rel.dtype and np.array_equal(o_rel, gt_rel)
    assert int((z[tag + "_count"] == [int((inst == i).sum()) for i in nodes]).all())      # np.where list lengths the draws came from


@pytest.mark.parametrize("chan", ["xyz", "xyz_rgb_normal"])
def test_oracle_prepare_objects_equals_data_preparation_on_the_recorded_draws(gold, chan):
    """obj_points [N,P,C] and descriptor [N,11] of data_preparation (:285-293) given the draws np.random.choice made there: the
    oracle's zero-mean and its float64 descriptor are EXACTLY the reference's; extra channels ride along uncentred."""
    z, e, d = gold
    rel, objs = _scan_a(gold)
    tag = "prep_multi_all_" + chan
    inst, pts = z["mesh_instances"], z["mesh_points_xyz_rgb_normal"]
    nodes = S.scene_nodes(inst, objs)
    choice = np.stack([np.where(inst == i)[0][z[tag + "_choice"][k]] for k, i in enumerate(nodes)])
    obj, desc = PO.prepare_objects(pts[:, :3], choice, torch.float64)
    want = z[tag + "_obj_points"]                                                       # [N,P,C]
    assert np.array_equal(obj.permute(0, 2, 1).numpy(), want[:, :, :3])
    assert np.array_equal(desc.numpy(), z[tag + "_descriptor"])
    if chan != "xyz":
        assert np.array_equal(pts[:, 3:][choice].astype(np.float32), want[:, :, 3:])
    else:
        assert np.array_equal(z[tag + "_obj_2d_feats"], z["multi_view_feats"])         # np.load of the multi-view files (:296-297)


def test_zero_mean_and_feature_paths(gold):
    z, e, d = gold
    assert np.array_equal(PO.zero_mean(torch.from_numpy(z["zero_mean_in"])).numpy(), z["zero_mean_out"])
    _, objs = _scan_a(gold)
    for i, rel_path in e["multi_view_relpaths"].items():
        assert S.multi_view_feature_path("/root_dir", "scan-a", int(i), objs[int(i)]) == os.path.join("/root_dir", rel_path)


def test_collate_equals_collate_fn_mmg(gold):
    """__getitem__ of two scans (5 nodes / 20 edges, 1 node / no edge) + collate_fn_mmg (DataLoader.py:153-176) vs synth.collate and
    the oracle's batching helper, after the layout change process_val / forward apply (model.py:76-82: [N,P,C] -> [N,C,P],
    [E,2] -> [2,E])."""
    z, e, d = gold
    items = []
    for k in (0, 1):
        items.append({"obj_points": z[f"item{k}_obj_points"].transpose(0, 2, 1), "obj_2d_feats": z[f"item{k}_obj_2d_feats"],
                      "edge_indices": z[f"item{k}_edge_indices"].reshape(-1, 2).T.astype(np.int64), "descriptor": z[f"item{k}_descriptor"]})
    got = synth.collate(items)
    assert np.array_equal(got["obj_points"], z["collate_obj_points"].transpose(0, 2, 1))
    assert np.array_equal(got["obj_2d_feats"], z["collate_obj_2d_feats"]) and np.array_equal(got["descriptor"], z["collate_descriptor"])
    assert np.array_equal(got["edge_indices"], z["collate_edge_indices"].T) and np.array_equal(got["batch_ids"], z["collate_batch_ids"])
    oe, ob = PO.fc_edges_batch([5, 1])
    assert np.array_equal(oe.numpy(), z["collate_edge_indices"]) and np.array_equal(ob.numpy(), z["collate_batch_ids"])
    # the second scan of the list the reference walked is scan-b_1: one annotated object, no edge, empty label rows
    assert e["getitem_scans"][:2] == ["scan-a_0", "scan-b_1"] and z["item1_edge_indices"].size == 0
    assert z["collate_gt_class"].shape == (6,) and z["collate_gt_rel_cls"].shape == (20, 26)
    nodes_b = S.scene_nodes(z["scan_b_instances"], {3: "bed"})
    assert nodes_b == [3] and S.edge_list(nodes_b, [], True).shape == (0, 2)
    gt_class_b, gt_rel_b = S.ground_truth(nodes_b, S.edge_list(nodes_b, [], True), {3: "bed"}, e["classes"], [], e["relations"], True)
    assert np.array_equal(gt_class_b, z["item1_gt_class"]) and gt_rel_b.shape == tuple(z["item1_gt_rels"].shape) == (0, 26)
