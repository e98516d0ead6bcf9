"""Real-data entry of the hot path (".ply -> tensors when 3RScan is available"): what the reference's dataset
class does on the host before ``Mmgnet.forward`` sees a scene, without trimesh --

* ``read_ply``                 the vertex element of a 3RScan label mesh (``labels.instances.align.annotated.v2.ply``): xyz, colours,
                               normals, instance id per vertex -- what ``load_mesh`` takes from trimesh (reference
                               ``src/dataset/dataset_3dssg.py:38-58``, ``utils/util_ply.py:8-14``);
* ``read_relationships``       ``relationships_*.json`` -> per-scan relationship lists and object-name maps
                               (``dataset_3dssg.py:215-243``, including the one scan the reference skips for the v2 label file);
* ``scene_nodes`` / ``edge_list`` / ``ground_truth``   the node order, the edge list and the labels of
                               ``data_preparation`` (``dataset_3dssg.py:248-266,281-283,300-336``);
* ``prepare_scan``             all of it + the per-object point selection, zero-mean, descriptor on the DEVICE (``prep.py``; reference
                               ``:279-294``) -> one batch dict in the layout ``evaluate.validation`` and ``VLSATModel.forward`` take.

The union point sets (``rel_points``, ``:337-359``) are not produced: ``Mmgnet.forward`` never reads them.  The host-side
functions are plain numpy; ``prepare_scan`` needs the GPU library.  Parity: PINNED to the reference's own dataset code --
``tests/golden/make_golden_scan.py`` imports ``dataset_3dssg`` / ``DataLoader`` / ``util`` / ``util_ply`` (with a stand-in for the
absent ``trimesh`` that only hands over the vertex table) and records what ``load_mesh``, ``read_relationship_json``,
``data_preparation`` (every combination of ``all_edge`` x ``multi_rel_outputs``, colour / normal channels, the draws of
``np.random.choice``), ``__getitem__`` + ``collate_fn_mmg`` and the name-list readers return; ``tests/test_scan_golden_cpu.py``
compares every function of this module with those fixtures, ``tests/test_hip_scan.py`` the device half on the recorded draws."""
from __future__ import annotations

import json
import os
from itertools import product
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
# the scan whose segments and ply mismatch in 3RScan v2 (reference dataset_3dssg.py:219-226)
_BAD_V2_SCAN = "fa79392f-7766-2d5c-869a-f5d6cfb62fc6"


class ScanError(ValueError):
    pass


def read_ply(path: str) -> Dict[str, Optional[np.ndarray]]:
    """-> {"points": f64[V,3], "colors": u8[V,3] | None, "normals": f64[V,3] | None, "instances": i64[V]}.
    ASCII and binary_little_endian PLY; only the vertex element is read (faces are skipped by never reaching them).  The instance id
    of a vertex is its ``objectId`` property, else ``label`` (reference utils/util_ply.py:8-14)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ScanError(f"{path}: not a PLY file")
        fmt, elements = None, []          # elements: [name, count, [(prop, dtype) | (prop, None) for lists]]
        while True:
            line = f.readline()
            if not line:
                raise ScanError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                if not elements:
                    raise ScanError(f"{path}: property before any element")
                if tok[1] == "list":
                    elements[-1][2].append((tok[-1], None))
                else:
                    if tok[1] not in _PLY_TYPES:
                        raise ScanError(f"{path}: unknown property type {tok[1]}")
                    elements[-1][2].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ScanError(f"{path}: format {fmt!r} is not supported (ascii, binary_little_endian)")
        vert = None
        for name, count, props in elements:
            scalar = all(t is not None for _, t in props)
            if name == "vertex":
                if not scalar:
                    raise ScanError(f"{path}: list property inside the vertex element")
                if fmt == "ascii":
                    vals = b" ".join(f.readline() for _ in range(count)).split()
                    if len(vals) != count * len(props):
                        raise ScanError(f"{path}: vertex rows do not hold {count} x {len(props)} values")
                    try:
                        table = np.array(vals, dtype=np.float64).reshape(count, len(props))
                    except ValueError:
                        raise ScanError(f"{path}: a vertex value is not a number") from None
                    # a value is parsed INTO its declared property type (a float32 written in decimal is that float32 again), as a
                    # PLY reader that fills the element's record does; integer columns go through int64 first
                    vert = {p: (table[:, i].astype(t) if t[0] == "f" else table[:, i].astype(np.int64).astype(t)) for i, (p, t) in enumerate(props)}
                else:
                    dt = np.dtype([(p, "<" + t) for p, t in props])
                    raw = f.read(dt.itemsize * count)
                    if len(raw) != dt.itemsize * count:
                        raise ScanError(f"{path}: vertex data truncated")
                    rec = np.frombuffer(raw, dtype=dt, count=count)
                    vert = {p: rec[p] for p, _ in props}
                break
            # an element in front of the vertices: skip it (scalar properties only -- a list would need parsing row by row)
            if not scalar:
                raise ScanError(f"{path}: element {name!r} with list properties precedes the vertices")
            if fmt == "ascii":
                for _ in range(count):
                    f.readline()
            else:
                f.seek(sum(np.dtype(t).itemsize for _, t in props) * count, os.SEEK_CUR)
        if vert is None:
            raise ScanError(f"{path}: no vertex element")
    for k in ("x", "y", "z"):
        if k not in vert:
            raise ScanError(f"{path}: vertex element has no {k}")
    out = {"points": np.stack([vert["x"], vert["y"], vert["z"]], 1).astype(np.float64)}
    out["colors"] = (np.stack([vert["red"], vert["green"], vert["blue"]], 1).astype(np.uint8)
                     if all(k in vert for k in ("red", "green", "blue")) else None)
    out["normals"] = (np.stack([vert["nx"], vert["ny"], vert["nz"]], 1).astype(np.float64)
                      if all(k in vert for k in ("nx", "ny", "nz")) else None)
    if "objectId" in vert:
        lab = vert["objectId"]
    elif "label" in vert:
        lab = vert["label"]
    else:
        raise ScanError(f"{path}: vertex element has neither objectId nor label")
    out["instances"] = np.asarray(lab).astype(np.int64).reshape(-1)
    return out


def scene_points(mesh: Dict[str, Optional[np.ndarray]], use_rgb: bool = False, use_normal: bool = False) -> np.ndarray:
    """xyz [+ rgb / 255] [+ normal] per vertex, the channel order of ``load_mesh`` (dataset_3dssg.py:43-52)."""
    pts = mesh["points"]
    if use_rgb:
        if mesh["colors"] is None:
            raise ScanError("USE_RGB: the mesh has no vertex colours")
        pts = np.concatenate([pts, mesh["colors"].astype(np.float64) / 255.0], 1)
    if use_normal:
        if mesh["normals"] is None:
            raise ScanError("USE_NORMAL: the mesh has no vertex normals")
        pts = np.concatenate([pts, mesh["normals"][:, :3]], 1)
    return pts


def read_name_list(path: str) -> List[str]:
    """classes.txt / relationships.txt: one name per line, exactly as the reference reads them (utils/util.py:15-21 ``read_txt_to_list``,
    :34-40 ``read_relationships``): trailing white space stripped, lower-cased, EVERY line kept -- an empty line is an entry (it
    takes an index), leading blanks stay."""
    with open(path, "r") as f:
        return [ln.rstrip().lower() for ln in f]


def read_relationships(path_or_data, selected_scans: Sequence[str],
                       label_file: str = "labels.instances.align.annotated.v2.ply") -> Tuple[dict, dict, list]:
    """relationships_{train,validation}.json -> (rel, objs, scans) keyed by ``<scan>_<split>`` exactly as the reference builds them
    (dataset_3dssg.py:215-243): rel[key] = list of [subject id, object id, relation id, relation name]; objs[key] = {instance id: label
    name} in the file's order (the node order of a scene follows it)."""
    data = path_or_data
    if isinstance(path_or_data, (str, os.PathLike)):
        with open(path_or_data) as f:
            data = json.load(f)
    selected = set(selected_scans)
    rel, objs, scans = {}, {}, []
    for scan_i in data["scans"]:
        if scan_i["scan"] == _BAD_V2_SCAN and label_file == "labels.instances.align.annotated.v2.ply":
            continue
        if scan_i["scan"] not in selected:
            continue
        key = scan_i["scan"] + "_" + str(scan_i["split"])
        rel[key] = [list(r) for r in scan_i["relationships"]]
        objs[key] = {int(i): name for i, name in scan_i["objects"].items()}
        scans.append(key)
    return rel, objs, scans


def scene_nodes(instances: np.ndarray, instance2label: Dict[int, str]) -> List[int]:
    """Instance ids of the scene's nodes, in the order of the object map, restricted to ids that own points; 0 is background
    (dataset_3dssg.py:251-261)."""
    present = set(int(i) for i in np.unique(instances))
    present.discard(0)
    return [int(i) for i in instance2label.keys() if int(i) in present]


def edge_list(nodes: Sequence[int], rel_json: Sequence[Sequence], all_edge: bool = True) -> np.ndarray:
    """i64[E,2] node-index pairs: every ordered pair i != j (source-major), or the annotated pairs only (dataset_3dssg.py:263-270)."""
    if all_edge:
        e = [(i, j) for i, j in product(range(len(nodes)), range(len(nodes))) if i != j]
    else:
        pos = {n: k for k, n in enumerate(nodes)}
        e = [(pos[r[0]], pos[r[1]]) for r in rel_json if r[0] in pos and r[1] in pos]
    return np.asarray(e, dtype=np.int64).reshape(-1, 2)


def ground_truth(nodes: Sequence[int], edges: np.ndarray, instance2label: Dict[int, str], class_names: Sequence[str],
                 rel_json: Sequence[Sequence], relation_names: Sequence[str], multi_rel_outputs: bool = True):
    """-> gt_class i64[N], gt_rel f32[E,R] (multi-label) | i64[E] (single label; 0 = none, a later annotation of a pair replaces
    an earlier one) -- dataset_3dssg.py:281-283,300-336.  Unknown names raise, as the reference's ``index`` / assert do."""
    class_pos = {n: k for k, n in enumerate(class_names)}
    rel_pos = {n: k for k, n in enumerate(relation_names)}
    try:
        gt_class = np.asarray([class_pos[instance2label[i]] for i in nodes], dtype=np.int64)
    except KeyError as e:
        raise ScanError(f"object label {e.args[0]!r} is not in the class list") from None
    pos = {n: k for k, n in enumerate(nodes)}
    n = len(nodes)
    adj = np.zeros((n, n, len(relation_names)), dtype=np.float32) if multi_rel_outputs else np.zeros((n, n), dtype=np.int64)
    for r in rel_json:
        if r[0] not in pos or r[1] not in pos:
            continue
        if r[3] not in rel_pos:
            raise ScanError(f"invalid relation name {r[3]!r}")
        k = rel_pos[r[3]]                  # (re-indexed by name: custom relation lists, :309)
        if multi_rel_outputs:
            adj[pos[r[0]], pos[r[1]], k] = 1.0
        else:
            adj[pos[r[0]], pos[r[1]]] = k
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    gt_rel = adj[edges[:, 0], edges[:, 1]] if len(edges) else np.zeros((0,) + adj.shape[2:], dtype=adj.dtype)   # (a one-object scan: no edge)
    return gt_class, gt_rel


def multi_view_feature_path(root: str, scene_id: str, instance_id: int, label: str) -> str:
    """Where the reference keeps an object's CLIP feature (dataset_3dssg.py:296-297)."""
    return os.path.join(root, f"data/3RScan/{scene_id}/multi_view/instance_{instance_id}_class_{label}_origin_view_mean.npy")


def prepare_scan(mesh_or_path, instance2label: Dict[int, str], class_names: Sequence[str], rel_json: Sequence[Sequence],
                 relation_names: Sequence[str], num_points: int, seed: int, device="cuda:0", multi_rel_outputs: bool = True,
                 all_edge: bool = True, use_rgb: bool = False, use_normal: bool = False, multi_view_root: Optional[str] = None,
                 scene_id: str = "", feature_loader: Optional[Callable[[int, str], np.ndarray]] = None) -> dict:
    """One scene, from the label mesh to the batch dict of ``evaluate.validation`` / ``VLSATModel.forward``: obj_points [N,C,P],
    obj_2d_feats [N,512], descriptor [N,11], edge_indices [E,2] (the loader's layout: ``forward`` takes its transpose, like the
    reference's ``process_val``), batch_ids [N,1], gt_class [N], gt_rel_cls [E,R] | [E], plus
    ``fc_sizes`` when the edge list is the fully connected one (and, for inspection, ``instance_ids``, ``points_per_instance``, ``choice``
    = the selected vertex indices [N,P]).  Points are selected, centred and described on the device
    (``prep.sample_objects`` / ``prep.prepare_objects``); extra channels (rgb, normals) are gathered with the same selection and are
    not centred (dataset_3dssg.py:291-293 centres xyz only).  ``feature_loader(instance id, label) -> f32[512]`` overrides the
    reference's file layout; without it and without ``multi_view_root`` the 2D features are zeros (as in the reference)."""
    import torch

    from . import prep

    mesh = read_ply(mesh_or_path) if isinstance(mesh_or_path, (str, os.PathLike)) else mesh_or_path
    nodes = scene_nodes(mesh["instances"], instance2label)
    if not nodes:
        raise ScanError("no annotated instance owns a point of this mesh")
    edges = edge_list(nodes, rel_json, all_edge)
    gt_class, gt_rel = ground_truth(nodes, edges, instance2label, class_names, rel_json, relation_names, multi_rel_outputs)
    pts = scene_points(mesh, use_rgb, use_normal)
    dev = torch.device(device)
    d_inst = torch.from_numpy(mesh["instances"].astype(np.int32)).to(dev)
    d_ids = torch.tensor(nodes, dtype=torch.int32, device=dev)
    if len(set(nodes)) != len(nodes) or min(nodes) < 0 or max(nodes) >= (1 << 24):
        raise ScanError("instance ids of the scene's nodes must be distinct integers in [0, 2^24)")
    choice, counts = prep.sample_objects(d_inst, d_ids, num_points, seed, map_size=max(65536, max(nodes) + 1))
    d_xyz = torch.from_numpy(np.ascontiguousarray(pts[:, :3], dtype=np.float32)).to(dev)
    obj_points, descriptor = prep.prepare_objects(d_xyz, choice)
    if pts.shape[1] > 3:                 # colour / normal channels ride along with the same selection
        extra = torch.from_numpy(np.ascontiguousarray(pts[:, 3:], dtype=np.float32)).to(dev)
        obj_points = torch.cat([obj_points, extra[choice.long()].permute(0, 2, 1).contiguous()], 1)
    feats = np.zeros((len(nodes), 512), dtype=np.float32)
    if feature_loader is not None or multi_view_root is not None:
        for k, i in enumerate(nodes):
            name = instance2label[i]
            feats[k] = (feature_loader(i, name) if feature_loader is not None
                        else np.load(multi_view_feature_path(multi_view_root, scene_id, i, name)))
    n = len(nodes)
    batch = {"obj_points": obj_points, "obj_2d_feats": torch.from_numpy(feats).to(dev), "descriptor": descriptor,
             "edge_indices": torch.from_numpy(edges).to(dev),
             "batch_ids": torch.zeros(n, 1, dtype=torch.int64, device=dev),
             "gt_class": torch.from_numpy(gt_class).to(dev), "gt_rel_cls": torch.from_numpy(gt_rel).to(dev),
             "n_scenes": 1, "instance_ids": nodes, "points_per_instance": counts, "choice": choice}
    if all_edge:
        batch["fc_sizes"] = [n]
    return batch
