#!/usr/bin/env python3
"""Golden vectors for the HOST half of the input preparation, made by the REAL reference dataset code:

    src/dataset/dataset_3dssg.py   load_mesh (:38-58), SSGDatasetGraph.read_relationship_json (:215-243), zero_mean (:189-191),
                                   data_preparation (:244-367), __getitem__ (:139-176)
    src/dataset/DataLoader.py      collate_fn_mmg (:153-176)
    utils/util_ply.py              read_labels (:8-14)
    utils/util.py                  read_txt_to_list (:15-21), read_relationships (:34-40)

Run once here:  python tests/golden/make_golden_scan.py       (the GPU box never sees /root/reference)

The dataset module imports `trimesh`, which this image lacks.  It is replaced by a stand-in with the one function the reference
calls (`trimesh.load(path, process=False)`), returning an object with the four attributes `load_mesh` / `read_labels` read --
`vertices`, `visual.vertex_colors` (RGBA), `vertex_normals`, `metadata['ply_raw']['vertex']['data']` -- filled from the PLY
file THIS script wrote, through the numpy record layout it wrote it with (no PLY parser of the product is involved).
`np.random.choice` is wrapped to RECORD what the reference drew, so that the tests can hand the product the same selection.

Written: scan_small.ply / scan_small_ascii.ply (the label mesh, both encodings), scan_small_relationships.json (annotation
document), scan_names_messy.txt (a names file with trailing blanks, capitals and an empty line), scan_small.npz (arrays) and
scan_small_expect.json (lists / dicts) -- inputs and the reference's outputs, nothing else.
"""
import copy
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402

LABEL_FILE = "labels.instances.align.annotated.v2.ply"
VERT = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"),
                 ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("objectId", "<u2"), ("globalId", "<u2"), ("NYU40", "u1")])
PLY_NAMES = {"<f4": "float", "u1": "uchar", "|u1": "uchar", "<u2": "ushort"}


def ply_header(n, binary):
    head = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"), "comment tests/golden/make_golden_scan.py",
            "element vertex %d" % n]
    head += ["property %s %s" % (PLY_NAMES[VERT[name].str], name) for name in VERT.names]
    head += ["element face 1", "property list uchar int vertex_indices", "end_header"]
    return ("\n".join(head) + "\n").encode()


def write_ply(path, rec, binary):
    with open(path, "wb") as f:
        f.write(ply_header(len(rec), binary))
        if binary:
            f.write(rec.tobytes())
            f.write(np.array([3], "u1").tobytes() + np.array([0, 1, 2], "<i4").tobytes())
        else:
            for r in rec:
                f.write((" ".join(("%.9g" % float(r[k])) if VERT[k].kind == "f" else str(int(r[k])) for k in VERT.names) + "\n").encode())
            f.write(b"3 0 1 2\n")


class _StubMesh:
    """What trimesh.load(..., process=False) exposes of a PLY label mesh, as far as the reference reads it."""

    def __init__(self, rec):
        self.vertices = np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float64)
        rgba = np.stack([rec["red"], rec["green"], rec["blue"], np.full(len(rec), 255, np.uint8)], 1)
        self.visual = types.SimpleNamespace(vertex_colors=rgba)
        self.vertex_normals = np.stack([rec["nx"], rec["ny"], rec["nz"]], 1).astype(np.float64)
        self.metadata = {"ply_raw": {"vertex": {"data": rec}}}


def install_trimesh_stub():
    def load(path, process=False, **kw):
        raw = open(path, "rb").read()
        body = raw[raw.index(b"end_header\n") + len(b"end_header\n"):]
        n = int([ln for ln in raw.split(b"\n") if ln.startswith(b"element vertex")][0].split()[-1])
        return _StubMesh(np.frombuffer(body, dtype=VERT, count=n))                 # binary files of this script only
    tm = types.ModuleType("trimesh")
    tm.load = load
    sys.modules["trimesh"] = tm


def scene(seed, n_pts, ids):
    g = np.random.default_rng(seed)
    rec = np.zeros(n_pts, dtype=VERT)
    inst = g.choice(np.array((0,) + tuple(ids)), n_pts)
    centres = {i: g.uniform(-2, 2, 3) for i in (0,) + tuple(ids)}
    xyz = np.stack([centres[int(i)] for i in inst]) + g.normal(size=(n_pts, 3)) * g.uniform(0.1, 0.6, (n_pts, 1))
    rec["x"], rec["y"], rec["z"] = xyz.astype(np.float32).T
    rec["red"], rec["green"], rec["blue"] = g.integers(0, 256, (3, n_pts))
    nrm = g.normal(size=(n_pts, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    rec["nx"], rec["ny"], rec["nz"] = nrm.astype(np.float32).T
    rec["objectId"] = inst
    rec["globalId"] = inst + 100
    rec["NYU40"] = 7
    return rec


def relationships_doc(bad_scan):
    return {"scans": [
        # object map in an order that is NOT the sorted one; 77 owns no vertex; vertices of 12 exist but 12 is not annotated here
        {"scan": "scan-a", "split": 0, "objects": {"5": "chair", "1": "floor", "9": "table", "77": "lamp", "2": "wall", "40": "sofa"},
         "relationships": [[5, 1, 14, "standing on"], [9, 1, 14, "standing on"], [5, 9, 3, "close by"], [5, 9, 7, "left"],
                           [77, 1, 14, "standing on"], [2, 1, 1, "attached to"], [40, 5, 3, "close by"], [5, 1, 14, "standing on"],
                           [1, 40, 15, "supported by"]]},
        {"scan": "scan-b", "split": 1, "objects": {"3": "bed"}, "relationships": []},
        {"scan": bad_scan, "split": 0, "objects": {"1": "floor"}, "relationships": []},
        {"scan": "scan-a", "split": 1, "objects": {"12": "sofa", "1": "floor", "2": "wall"},
         "relationships": [[12, 1, 14, "standing on"], [2, 1, 1, "attached to"]]},
    ]}


def main():
    G.install_standins()
    install_trimesh_stub()
    from src.dataset import dataset_3dssg as D
    from src.dataset.DataLoader import collate_fn_mmg
    from utils import util, util_ply

    expect, arrays = {}, {}
    # ---- name lists: the reference's readers on the subset's files and on a messy file
    messy = os.path.join(HERE, "scan_names_messy.txt")
    open(messy, "w").write("Chair  \nFLOOR\n\ntable\t\n  lamp\nWall\r\nsofa")
    expect["messy_read_txt_to_list"] = util.read_txt_to_list(messy)
    expect["messy_read_relationships"] = util.read_relationships(messy)
    classes = util.read_txt_to_list(os.path.join(HERE, "3dssg_classes.txt"))
    relations_all = util.read_relationships(os.path.join(HERE, "3dssg_relations.txt"))
    expect["classes"], expect["relations"] = classes, relations_all       # (this list is the 26 names without 'none')

    # ---- the label mesh, written here, read by the reference's load_mesh through the trimesh stand-in
    rec = scene(21, 900, ids=(1, 2, 5, 9, 12, 40))
    tmp = tempfile.mkdtemp(prefix="vlsat_scan_golden_")
    scan_dir = os.path.join(tmp, "3RScan", "scan-a")
    os.makedirs(scan_dir)
    write_ply(os.path.join(scan_dir, LABEL_FILE), rec, True)
    shutil.copy(os.path.join(scan_dir, LABEL_FILE), os.path.join(HERE, "scan_small.ply"))
    write_ply(os.path.join(HERE, "scan_small_ascii.ply"), rec, False)
    m9 = D.load_mesh(scan_dir, LABEL_FILE, True, True)
    m3 = D.load_mesh(scan_dir, LABEL_FILE, False, False)
    m6n = D.load_mesh(scan_dir, LABEL_FILE, False, True)
    assert np.array_equal(m9["points"][:, :3], m3["points"]) and np.array_equal(m9["points"][:, 6:], m6n["points"][:, 3:])
    arrays["mesh_points_xyz_rgb_normal"] = m9["points"]
    arrays["mesh_instances"] = np.asarray(m9["instances"]).astype(np.int64)
    # read_labels prefers objectId, falls back to label
    only_label = np.zeros(4, dtype=[("x", "<f4"), ("label", "<u2")])
    only_label["label"] = [4, 0, 9, 9]
    stub = types.SimpleNamespace(metadata={"ply_raw": {"vertex": {"data": only_label}}})
    expect["read_labels_fallback"] = [int(v) for v in util_ply.read_labels(stub)]

    # ---- relationships json
    ds = object.__new__(D.SSGDatasetGraph)
    bad = "fa79392f-7766-2d5c-869a-f5d6cfb62fc6"
    doc = relationships_doc(bad)
    json.dump(doc, open(os.path.join(HERE, "scan_small_relationships.json"), "w"), indent=1)
    for tag, label_file in (("v2", LABEL_FILE), ("v1", "labels.instances.align.annotated.ply")):
        ds.mconfig = types.SimpleNamespace(label_file=label_file)
        rel, objs, scans = ds.read_relationship_json(copy.deepcopy(doc), ["scan-a", bad])
        expect["relationships_" + tag] = {"rel": rel, "objs": {k: [[i, n] for i, n in v.items()] for k, v in objs.items()}, "scans": scans}
    ds.mconfig = types.SimpleNamespace(label_file=LABEL_FILE)
    rel, objs, scans = ds.read_relationship_json(copy.deepcopy(doc), ["scan-a", bad])

    # ---- zero_mean on its own
    g = np.random.default_rng(5)
    zm_in = g.normal(size=(7, 3)).astype(np.float32) * 3 + 1
    arrays["zero_mean_in"] = zm_in
    arrays["zero_mean_out"] = ds.zero_mean(torch.from_numpy(zm_in.copy())).numpy()

    # ---- data_preparation, every switch that changes its outputs; np.random.choice recorded
    P, PU = 16, 8
    drawn = []
    real_choice = np.random.choice

    def recording_choice(a, size=None, replace=True, p=None):
        out = real_choice(a, size, replace, p)
        drawn.append((int(a), int(size), np.asarray(out).copy()))
        return out

    np.random.choice = recording_choice
    key = "scan-a_0"
    feat_root = os.path.join(tmp, "mv")
    nodes_expected = [5, 1, 9, 2, 40]
    fg = np.random.default_rng(9)
    feats = {}
    for i in nodes_expected:
        p = os.path.join(feat_root, f"data/3RScan/scan-a/multi_view/instance_{i}_class_{objs[key][i]}_origin_view_mean.npy")
        os.makedirs(os.path.dirname(p), exist_ok=True)
        feats[i] = fg.normal(size=512)
        np.save(p, feats[i])
    expect["multi_view_relpaths"] = {str(i): os.path.relpath(
        os.path.join(feat_root, f"data/3RScan/scan-a/multi_view/instance_{i}_class_{objs[key][i]}_origin_view_mean.npy"), feat_root)
        for i in nodes_expected}
    try:
        for multi in (True, False):
            names = relations_all if multi else ["none"] + relations_all        # (multi-label drops 'none': dataset_3dssg.py:95-96)
            for all_edge in (True, False):
                for chan, pts in (("xyz", m3["points"]), ("xyz_rgb_normal", m9["points"])):
                    if chan != "xyz" and not (multi and all_edge):
                        continue
                    np.random.seed(1234)
                    drawn.clear()
                    out = ds.data_preparation(pts.copy(), m9["instances"], P, PU, scene_id="scan-a", instance2labelName=objs[key],
                                              classNames=classes, rel_json=copy.deepcopy(rel[key]), relationships=list(names),
                                              multi_rel_outputs=multi, all_edge=all_edge,
                                              multi_view_root=feat_root if chan == "xyz" and multi and all_edge else None)
                    obj_points, obj_2d, rel_points, gt_rels, label_node, edge_indices, descriptor = out
                    tag = f"prep_{'multi' if multi else 'single'}_{'all' if all_edge else 'annot'}_{chan}"
                    n = obj_points.shape[0]
                    arrays[tag + "_choice"] = np.stack([d[2] for d in drawn[:n]]).astype(np.int64)      # draws WITHIN each object's list
                    arrays[tag + "_count"] = np.asarray([d[0] for d in drawn[:n]], dtype=np.int64)
                    assert all(d[1] == P for d in drawn[:n]) and all(d[1] == PU for d in drawn[n:])
                    arrays[tag + "_obj_points"] = obj_points.numpy()
                    arrays[tag + "_obj_2d_feats"] = obj_2d.numpy()
                    arrays[tag + "_gt_rels"] = gt_rels.numpy()
                    arrays[tag + "_label_node"] = label_node.numpy()
                    arrays[tag + "_edge_indices"] = edge_indices.numpy().reshape(-1, 2)
                    arrays[tag + "_descriptor"] = descriptor.numpy()
                    expect[tag + "_rel_points_shape"] = list(rel_points.shape)
        # ---- __getitem__ of two scans + collate_fn_mmg (what the DataLoader hands to process_val)
        os.makedirs(os.path.join(tmp, "3RScan", "scan-b"))
        rec_b = scene(22, 300, ids=(3, 4))
        write_ply(os.path.join(tmp, "3RScan", "scan-b", LABEL_FILE), rec_b, True)
        rel2, objs2, scans2 = ds.read_relationship_json(copy.deepcopy(doc), ["scan-a", "scan-b"])
        ds.scans, ds.objs_json, ds.relationship_json = scans2, objs2, rel2
        ds.root_3rscan = os.path.join(tmp, "3RScan")
        ds.mconfig = types.SimpleNamespace(label_file=LABEL_FILE, num_points=P, num_points_union=PU)
        ds.use_rgb = ds.use_normal = ds.for_train = False
        ds.classNames, ds.relationNames, ds.multi_rel_outputs = classes, relations_all, True
        ds.max_edges, ds.shuffle_objs, ds.use_2d_feats, ds.use_descriptor = -1, False, True, True
        ds.config = types.SimpleNamespace(multi_view_root=None)
        expect["getitem_scans"] = scans2
        items = []
        for idx in (0, 1):                                             # scan-a_0 (5 nodes), scan-b_1 (1 node: no edge)
            np.random.seed(77 + idx)
            drawn.clear()
            it = ds[idx]
            n = it[0].shape[0]
            arrays[f"item{idx}_choice"] = np.stack([d[2] for d in drawn[:n]]).astype(np.int64)
            items.append(it)
        arrays["scan_b_instances"] = np.asarray(rec_b["objectId"]).astype(np.int64)
        arrays["scan_b_points"] = np.stack([rec_b["x"], rec_b["y"], rec_b["z"]], 1).astype(np.float64)
        col = collate_fn_mmg(items)
        for name, t in zip(("obj_points", "obj_2d_feats", "gt_class", "gt_rel_cls", "edge_indices", "descriptor", "batch_ids"), col):
            arrays["collate_" + name] = t.numpy()
        for idx, it in enumerate(items):
            for name, t in zip(("obj_points", "obj_2d_feats", "rel_points", "gt_class", "gt_rels", "edge_indices", "descriptor"), it):
                if name != "rel_points":
                    arrays[f"item{idx}_{name}"] = t.numpy()
    finally:
        np.random.choice = real_choice
        shutil.rmtree(tmp, ignore_errors=True)
    arrays["multi_view_feats"] = np.stack([feats[i] for i in nodes_expected]).astype(np.float32)
    expect["nodes_scan_a_0"] = nodes_expected
    np.savez_compressed(os.path.join(HERE, "scan_small.npz"), **arrays)
    json.dump(expect, open(os.path.join(HERE, "scan_small_expect.json"), "w"), indent=1, sort_keys=True)
    print("written", len(arrays), "arrays;", {k: v.shape for k, v in arrays.items() if k.startswith("prep_multi_all_xyz_")})


if __name__ == "__main__":
    main()
