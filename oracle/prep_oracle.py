"""CPU ORACLE for the per-object input preparation  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates what the reference data loader does per object after sampling:
  descriptor = gen_descriptor(sampled raw points)            reference src/utils/op_utils.py:47-64
  points     = zero_mean(sampled points as float32)          reference src/dataset/dataset_3dssg.py:189-191,289-293
  layout     = permute [N,P,3] -> [N,3,P]                    reference src/model/model.py:79
  FC edges   = product(range(n), range(n)) minus diagonal    reference src/dataset/dataset_3dssg.py:264-266
  batching   = node offsets + batch_ids                       reference src/dataset/DataLoader.py:160-172
Parity status: PINNED.  gen_descriptor by tests/test_prep_oracle.py against tests/golden/prep_small.npz (reference function called
directly); zero_mean, node order, edge list, labels, the [N,P,C] object tensors, the descriptor as data_preparation stores it
(float64 arithmetic on trimesh's float64 vertices, cast to float32) and collate_fn_mmg by tests/test_scan_golden_cpu.py against
tests/golden/scan_small.npz, which tests/golden/make_golden_scan.py makes by running the reference's dataset module itself
(trimesh, absent here, replaced by a stand-in that hands over the vertex table; np.random.choice recorded).
The random sampling itself (np.random.choice, :289) is an input of prepare_objects (`choice`); sample_choice restates the
library's OWN documented generator for the device-side selection (vlsat_sample_objects): that one has no reference
counterpart to pin to beyond np.where's index order -- the tests check membership, uniformity and determinism."""
from __future__ import annotations

import numpy as np
import torch


def gen_descriptor(pts: torch.Tensor) -> torch.Tensor:
    """[P,3] -> [11]: centroid, unbiased std, dims, volume, max dim (op_utils.py:47-64)."""
    dims = pts.max(dim=0)[0] - pts.min(dim=0)[0]
    return torch.cat([pts.mean(0), pts.std(0), dims, (dims[0] * dims[1] * dims[2]).unsqueeze(0), dims.max().unsqueeze(0)], 0)


def zero_mean(point: torch.Tensor) -> torch.Tensor:
    """[P,3] -> centred copy (dataset_3dssg.py:189-191; the reference centres in place)."""
    return point - torch.mean(point, dim=0).unsqueeze(0)


def prepare_objects(scene_points: np.ndarray, choice: np.ndarray, dtype=torch.float32):
    """scene_points [Npts,3], choice [N,P] (sampled point ids) -> obj_points [N,3,P] f32, descriptor [N,11] f32.  ``dtype`` is the
    arithmetic of the descriptor: the reference computes it on the mesh's vertices as trimesh returns them, float64, and stores the
    result in a float32 tensor (dataset_3dssg.py:290, 273) -- torch.float64 with float64 ``scene_points`` reproduces that exactly."""
    n, p = choice.shape
    obj = torch.zeros(n, p, 3)
    desc = torch.zeros(n, 11)
    for i in range(n):
        pts = torch.from_numpy(scene_points[choice[i]])
        desc[i] = gen_descriptor(pts.to(dtype))
        obj[i] = zero_mean(pts.to(torch.float32))
    return obj.permute(0, 2, 1).contiguous(), desc


def sample_choice(instances: np.ndarray, instance_ids, n_sample: int, seed: int):
    """The point selection of dataset_3dssg.py:285-289 with the library's documented generator (include/vlsat.h,
    vlsat_sample_objects) instead of np.random: per object the ascending index list np.where(instances == id)[0], and draw j =
    list[((splitmix64(seed + 0x9E3779B97F4A7C15 (obj n_sample + j + 1)) >> 32) * len) >> 32].  -> choice [N, n_sample], counts [N]."""
    M = (1 << 64) - 1
    choice = np.zeros((len(instance_ids), n_sample), dtype=np.int64)
    counts = np.zeros(len(instance_ids), dtype=np.int64)
    for o, iid in enumerate(instance_ids):
        lst = np.where(instances == iid)[0]
        counts[o] = len(lst)
        for j in range(n_sample):
            z = (seed + 0x9E3779B97F4A7C15 * (o * n_sample + j + 1)) & M
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            z ^= z >> 31
            if len(lst):
                choice[o, j] = lst[((z >> 32) * len(lst)) >> 32]
    return choice, counts


def fc_edges_batch(n_per_scene):
    """-> edge_indices [E,2] int64 (from, to), batch_ids [N,1] int64, as collate_fn_mmg yields them."""
    edges, bids, off = [], [], 0
    for s, n in enumerate(n_per_scene):
        e = [(i + off, j + off) for i in range(n) for j in range(n) if i != j]
        edges.append(torch.tensor(e, dtype=torch.long).view(-1, 2))
        bids.append(torch.full((n, 1), s, dtype=torch.long))
        off += n
    return torch.cat(edges, 0), torch.cat(bids, 0)


def scene_labels(instances: np.ndarray, instance2label: dict, class_names, rel_json, relation_names, multi_rel_outputs=True, all_edge=True):
    """Node order, edge list and ground truth of one scene, loop by loop as the reference's data_preparation builds them
    (src/dataset/dataset_3dssg.py:248-270 nodes and edges, :281-283 object labels, :300-314 adjacency, :322-336 per-edge labels).
    Pinned by tests/test_scan_golden_cpu.py to what the reference's data_preparation returns.  -> nodes, edges [E,2], gt_class [N], gt_rel."""
    ids_with_points = list(np.unique(instances))
    if 0 in ids_with_points:
        ids_with_points.remove(0)                                   # background
    nodes = []
    for inst in list(instance2label.keys()):
        if inst in ids_with_points:
            nodes.append(inst)
    if all_edge:
        edges = []
        for i in range(len(nodes)):
            for j in range(len(nodes)):
                if i != j:
                    edges.append((i, j))
    else:
        edges = [(nodes.index(r[0]), nodes.index(r[1])) for r in rel_json if r[0] in nodes and r[1] in nodes]
    gt_class = [list(class_names).index(instance2label[inst]) for inst in nodes]
    n, R = len(nodes), len(relation_names)
    adj = np.zeros([n, n, R]) if multi_rel_outputs else np.zeros([n, n])
    for r in rel_json:
        if r[0] not in nodes or r[1] not in nodes:
            continue
        assert r[3] in relation_names
        k = list(relation_names).index(r[3])
        if multi_rel_outputs:
            adj[nodes.index(r[0]), nodes.index(r[1]), k] = 1
        else:
            adj[nodes.index(r[0]), nodes.index(r[1])] = k
    gt = np.zeros([len(edges), R], dtype=np.float32) if multi_rel_outputs else np.zeros([len(edges)], dtype=np.int64)
    for e, (a, b) in enumerate(edges):
        gt[e] = adj[a, b]
    return nodes, np.asarray(edges, dtype=np.int64).reshape(-1, 2), np.asarray(gt_class, dtype=np.int64), gt
