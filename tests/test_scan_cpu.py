"""Host side of the real-data entry (vlsat_amd/scan.py): PLY vertex reader, relationships json, node order / edge list / labels
against the loop-by-loop restatement in oracle/prep_oracle.py.  CPU only."""
import json
import os
import struct

import numpy as np
import pytest

import vlsat_amd  # noqa: F401
from vlsat_amd import scan as S
from oracle import prep_oracle as PO

PROPS = [("x", "float", "f"), ("y", "float", "f"), ("z", "float", "f"), ("red", "uchar", "B"), ("green", "uchar", "B"), ("blue", "uchar", "B"),
         ("objectId", "ushort", "H"), ("globalId", "ushort", "H"), ("NYU40", "uchar", "B"), ("Eigen13", "uchar", "B"), ("RIO27", "uchar", "B")]


def write_ply(path, pts, rgb, inst, binary, label_name="objectId", with_faces=True, lead_element=False):
    """A label mesh with the property list of 3RScan's labels.instances.align.annotated.v2.ply (+ optionally an element in front)."""
    n = len(pts)
    props = [(label_name if p == "objectId" else p, t, c) for p, t, c in PROPS]
    head = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"), "comment written by tests/test_scan_cpu.py"]
    if lead_element:
        head += ["element camera 2", "property float fx", "property int id"]
    head += ["element vertex %d" % n] + ["property %s %s" % (t, p) for p, t, _ in props]
    if with_faces:
        head += ["element face 2", "property list uchar int vertex_indices"]
    head += ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if lead_element:
            for k in range(2):
                f.write(struct.pack("<fi", 1.5 + k, k) if binary else ("%g %d\n" % (1.5 + k, k)).encode())
        for i in range(n):
            row = [*pts[i], *rgb[i], inst[i], inst[i] + 100, 1, 2, 3]
            if binary:
                f.write(struct.pack("<" + "".join(c for _, _, c in props), *[float(v) if c == "f" else int(v) for v, (_, _, c) in zip(row, props)]))
            else:
                f.write((" ".join(repr(float(v)) if c == "f" else str(int(v)) for v, (_, _, c) in zip(row, props)) + "\n").encode())
        if with_faces:
            for _ in range(2):
                f.write(struct.pack("<Biii", 3, 0, 1, 2) if binary else b"3 0 1 2\n")


def make_scene(seed=0, n_pts=500, ids=(1, 2, 5, 9, 12)):
    g = np.random.default_rng(seed)
    pts = g.normal(size=(n_pts, 3)).astype(np.float32)
    rgb = g.integers(0, 256, (n_pts, 3))
    inst = g.choice(np.array((0,) + tuple(ids)), n_pts)
    return pts, rgb, inst


@pytest.mark.parametrize("binary", [False, True])
@pytest.mark.parametrize("lead", [False, True])
def test_read_ply_returns_what_was_written(tmp_path, binary, lead):
    pts, rgb, inst = make_scene()
    p = str(tmp_path / "labels.instances.align.annotated.v2.ply")
    write_ply(p, pts, rgb, inst, binary, lead_element=lead)
    m = S.read_ply(p)
    assert m["points"].dtype == np.float64 and np.array_equal(m["points"].astype(np.float32), pts)
    assert np.array_equal(m["colors"], rgb.astype(np.uint8)) and m["normals"] is None
    assert m["instances"].dtype == np.int64 and np.array_equal(m["instances"], inst)
    six = S.scene_points(m, use_rgb=True)
    assert six.shape == (len(pts), 6) and np.allclose(six[:, 3:], rgb / 255.0)
    with pytest.raises(S.ScanError):
        S.scene_points(m, use_normal=True)


def test_read_ply_label_property_and_errors(tmp_path):
    pts, rgb, inst = make_scene(1, 40)
    p = str(tmp_path / "a.ply")
    write_ply(p, pts, rgb, inst, True, label_name="label", with_faces=False)
    assert np.array_equal(S.read_ply(p)["instances"], inst)            # no objectId: `label` (util_ply.read_labels)
    write_ply(p, pts, rgb, inst, True, label_name="segment")
    with pytest.raises(S.ScanError, match="objectId"):
        S.read_ply(p)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(S.ScanError, match="truncated"):
        S.read_ply(p)
    open(p, "wb").write(b"plx\n")
    with pytest.raises(S.ScanError, match="not a PLY"):
        S.read_ply(p)
    open(p, "wb").write(b"ply\nformat binary_big_endian 1.0\nelement vertex 0\nproperty float x\nend_header\n")
    with pytest.raises(S.ScanError, match="not supported"):
        S.read_ply(p)


def relationships_doc():
    return {"scans": [
        {"scan": "s-a", "split": 0, "objects": {"5": "chair", "1": "floor", "9": "table", "77": "lamp", "2": "wall"},
         "relationships": [[5, 1, 14, "standing on"], [9, 1, 14, "standing on"], [5, 9, 3, "close by"], [5, 9, 7, "left"], [77, 1, 14, "standing on"],
                           [2, 1, 1, "attached to"]]},
        {"scan": "s-b", "split": 1, "objects": {"3": "bed"}, "relationships": []},
        {"scan": S._BAD_V2_SCAN, "split": 0, "objects": {"1": "floor"}, "relationships": []},
        {"scan": "s-a", "split": 1, "objects": {"12": "sofa", "1": "floor"}, "relationships": [[12, 1, 14, "standing on"]]},
    ]}


def test_read_relationships_keys_order_and_the_skipped_scan(tmp_path):
    p = str(tmp_path / "relationships_validation.json")
    json.dump(relationships_doc(), open(p, "w"))
    rel, objs, scans = S.read_relationships(p, ["s-a", S._BAD_V2_SCAN])
    assert scans == ["s-a_0", "s-a_1"] and set(rel) == set(objs) == set(scans)
    assert list(objs["s-a_0"].keys()) == [5, 1, 9, 77, 2] and objs["s-a_0"][9] == "table"           # int keys, file order
    assert rel["s-a_0"][2] == [5, 9, 3, "close by"]
    _, _, scans_v1 = S.read_relationships(relationships_doc(), ["s-a", S._BAD_V2_SCAN], label_file="labels.instances.align.annotated.ply")
    assert S._BAD_V2_SCAN + "_0" in scans_v1                                                          # only the v2 label file drops it


@pytest.mark.parametrize("multi", [True, False])
@pytest.mark.parametrize("all_edge", [True, False])
def test_nodes_edges_labels_equal_the_restatement(multi, all_edge):
    doc = relationships_doc()["scans"][0]
    objs = {int(k): v for k, v in doc["objects"].items()}
    _, _, inst = make_scene(3, 800, ids=(1, 2, 5, 9, 12))          # 77 owns no point, 12 is not annotated
    classes = ["bed", "chair", "floor", "lamp", "sofa", "table", "wall"]
    rels = ["none", "attached to", "close by", "left", "standing on"] if not multi else ["attached to", "close by", "left", "standing on"]
    nodes = S.scene_nodes(inst, objs)
    edges = S.edge_list(nodes, doc["relationships"], all_edge)
    gt_class, gt_rel = S.ground_truth(nodes, edges, objs, classes, doc["relationships"], rels, multi)
    r_nodes, r_edges, r_class, r_rel = PO.scene_labels(inst, objs, classes, doc["relationships"], rels, multi, all_edge)
    assert nodes == r_nodes == [5, 1, 9, 2]
    assert np.array_equal(edges, r_edges) and np.array_equal(gt_class, r_class)
    assert gt_rel.dtype == r_rel.dtype and np.array_equal(gt_rel, r_rel)
    if all_edge:
        assert len(edges) == 12 and tuple(edges[0]) == (0, 1) and tuple(edges[3]) == (1, 0)
    if multi:
        e = [tuple(x) for x in edges].index((0, 2))
        assert gt_rel[e].tolist() == [0, 1, 1, 0]                   # chair -> table: close by AND left
    else:
        e = [tuple(x) for x in edges].index((0, 2))
        assert gt_rel[e] == 3                                       # the later annotation of a pair replaces the earlier one
    with pytest.raises(S.ScanError, match="relation"):
        S.ground_truth(nodes, edges, objs, classes, [[5, 1, 0, "hovering over"]], rels, multi)
    with pytest.raises(S.ScanError, match="class list"):
        S.ground_truth(nodes, edges, objs, [c for c in classes if c != "chair"], doc["relationships"], rels, multi)


def test_name_lists_of_the_reference_subset_have_the_survey_sizes():
    """classes.txt / relations.txt as 3DSSG ships them: 160 classes, 26 relations + none (C = 160, R = 26).  The two lists
    are data files of the dataset; tests/golden keeps copies so that the test does not read /root/reference."""
    here = os.path.join(os.path.dirname(__file__), "golden")
    classes = S.read_name_list(os.path.join(here, "3dssg_classes.txt"))
    relations = S.read_name_list(os.path.join(here, "3dssg_relations.txt"))
    assert len(classes) == 160 and len(set(classes)) == 160
    assert len(relations) in (26, 27)


def test_read_ply_round_trips_random_vertex_tables(tmp_path):
    """Property test (hypothesis): any vertex table with the 3RScan label-mesh properties in any order, extra scalar properties of any
    PLY type, ASCII or binary, with or without an element in front -- read_ply returns exactly the written xyz / colours / normals /
    instance ids (values are parsed into their declared type, so a float32 written in decimal comes back as that float32)."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    types = {"float": ("<f4", "f"), "double": ("<f8", "d"), "uchar": ("u1", "B"), "ushort": ("<u2", "H"), "int": ("<i4", "i"), "short": ("<i2", "h")}

    @settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
    @given(st.integers(0, 2 ** 31 - 1), st.integers(1, 60), st.booleans(), st.booleans(), st.booleans(), st.booleans(), st.sampled_from(["objectId", "label"]),
           st.lists(st.sampled_from(sorted(types)), min_size=0, max_size=3))
    def run(seed, n, binary, lead, with_rgb, with_normals, label_name, extra):
        g = np.random.default_rng(seed)
        cols = [("x", "float"), ("y", "float"), ("z", "float")]
        if with_rgb:
            cols += [("red", "uchar"), ("green", "uchar"), ("blue", "uchar")]
        if with_normals:
            cols += [("nx", "float"), ("ny", "float"), ("nz", "float")]
        cols += [(label_name, "ushort")] + [(f"extra{i}", t) for i, t in enumerate(extra)]
        order = g.permutation(len(cols))
        cols = [cols[i] for i in order]
        rec = np.zeros(n, dtype=[(c, types[t][0]) for c, t in cols])
        for c, t in cols:
            kind = np.dtype(types[t][0]).kind
            rec[c] = g.normal(size=n) * 3 if kind == "f" else g.integers(0, 200 if t != "short" else 100, n)
        head = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii")]
        if lead:
            head += ["element camera 1", "property float fx"]
        head += ["element vertex %d" % n] + ["property %s %s" % (t, c) for c, t in cols] + ["element face 0", "property list uchar int vertex_indices", "end_header"]
        p = str(tmp_path / f"h_{seed}_{n}.ply")
        with open(p, "wb") as f:
            f.write(("\n".join(head) + "\n").encode())
            if lead:
                f.write(struct.pack("<f", 1.0) if binary else b"1.0\n")
            if binary:
                f.write(rec.tobytes())
            else:
                for r in rec:
                    f.write((" ".join(("%.9g" % float(r[c])) if np.dtype(types[t][0]).kind == "f" and t == "float" else repr(float(r[c])) if t == "double" else str(int(r[c])) for c, t in cols) + "\n").encode())
        m = S.read_ply(p)
        assert np.array_equal(m["points"], np.stack([rec["x"], rec["y"], rec["z"]], 1).astype(np.float64))
        assert np.array_equal(m["instances"], rec[label_name].astype(np.int64))
        if with_rgb:
            assert np.array_equal(m["colors"], np.stack([rec["red"], rec["green"], rec["blue"]], 1))
        else:
            assert m["colors"] is None
        if with_normals:
            assert np.array_equal(m["normals"], np.stack([rec["nx"], rec["ny"], rec["nz"]], 1).astype(np.float64))
        else:
            assert m["normals"] is None
    run()


def test_edge_list_and_ground_truth_agree_with_the_restatement_on_random_scans():
    """Property test: random object maps (any key order, instances without points, points without annotation), random relationship
    lists (duplicates, unknown instances, several labels per pair) -- scan.py's vectorised functions equal the oracle's loop-by-loop
    restatement of data_preparation (which tests/test_scan_golden_cpu.py pins to the reference itself), in all four switch settings."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=60, deadline=None)
    @given(st.integers(0, 2 ** 31 - 1), st.booleans(), st.booleans())
    def run(seed, multi, all_edge):
        g = np.random.default_rng(seed)
        ids = [int(i) for i in g.permutation(np.arange(1, 30))[: int(g.integers(1, 12))]]
        objs = {i: ["bed", "chair", "floor"][int(g.integers(0, 3))] for i in ids}
        with_points = [i for i in ids if g.random() < 0.8] + [int(g.integers(40, 50))]
        inst = g.choice(np.array([0] + with_points), 300)
        names = ["none", "left", "right", "on"] if not multi else ["left", "right", "on"]
        pool = ids + [99]
        rel = [[int(g.choice(pool)), int(g.choice(pool)), 0, str(g.choice(names[(0 if multi else 1):]))] for _ in range(int(g.integers(0, 25)))]
        nodes = S.scene_nodes(inst, objs)
        if not nodes:
            return
        edges = S.edge_list(nodes, rel, all_edge)
        gt_class, gt_rel = S.ground_truth(nodes, edges, objs, ["bed", "chair", "floor"], rel, names, multi)
        r_nodes, r_edges, r_class, r_rel = PO.scene_labels(inst, objs, ["bed", "chair", "floor"], rel, names, multi, all_edge)
        assert nodes == r_nodes and np.array_equal(edges, r_edges) and np.array_equal(gt_class, r_class)
        assert gt_rel.dtype == r_rel.dtype and gt_rel.shape == r_rel.shape and np.array_equal(gt_rel, r_rel)
    run()
