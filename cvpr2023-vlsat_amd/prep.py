"""Input preparation on the GPU: host-side mirror of the per-object part of
``SSGDatasetGraph.data_preparation`` (reference ``src/dataset/dataset_3dssg.py:279-294``), of the
fully-connected edge list (``:264-266``) and of ``collate_fn_mmg`` (``src/dataset/DataLoader.py:153-176``).
The selection of an object's points (``np.where(instances == id)`` + ``np.random.choice``, ``:285-289``) runs on the device too
(``sample_objects``): its index lists are np.where's, its draws come from a documented counter-based generator, not numpy's."""
from __future__ import annotations

from typing import Sequence

import torch

from . import lib as L


def sample_objects(instances: torch.Tensor, instance_ids: torch.Tensor, n_sample: int, seed: int, map_size: int = 65536):
    """instances i32[Npts] (instance id per scene point), instance_ids i32[N] (distinct) -> choice i32[N, n_sample] (indices into the
    scene's points, drawn with replacement from each instance's own points), counts i32[N] (points per instance).  Device tensors
    in and out; nothing is read back.  ``choice`` feeds ``prepare_objects``.  ``map_size`` bounds the instance ids the kernel can
    see: a requested id >= map_size gets count 0 and choice 0, and of two equal ids only one gets points -- callers that hold the
    ids on the host (``scan.prepare_scan``) check both and size the map from the largest id."""
    lib = L.load()
    dev = instances.device
    instances = instances.to(torch.int32).contiguous().view(-1)
    ids = instance_ids.to(device=dev, dtype=torch.int32).contiguous().view(-1)
    n_pts, n = instances.numel(), ids.numel()
    map_size = int(map_size)              # instance ids are small integers (3RScan: < 1000)
    if map_size <= 0:
        raise L.VlsatError("sample_objects: map_size must be positive")
    id_map = torch.empty(map_size, dtype=torch.int32, device=dev)
    scratch = torch.empty(int(lib.vlsat_sample_objects_scratch(n_pts, n)), dtype=torch.int32, device=dev)
    choice = torch.empty(n, int(n_sample), dtype=torch.int32, device=dev)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    L.check(lib.vlsat_sample_objects(instances.data_ptr(), n_pts, ids.data_ptr(), n, int(n_sample), int(seed) & (2 ** 64 - 1), id_map.data_ptr(),
                                     map_size, scratch.data_ptr(), choice.data_ptr(), counts.data_ptr(), L.stream_ptr()))
    return choice, counts


def prepare_objects(scene_points: torch.Tensor, choice: torch.Tensor):
    """scene_points f32[Npts,3], choice i32[N,P] (device) -> obj_points f32[N,3,P], descriptor f32[N,11]."""
    lib = L.load()
    scene_points = scene_points.contiguous()
    choice = choice.to(torch.int32).contiguous()
    if scene_points.dim() != 2 or scene_points.shape[1] != 3 or scene_points.dtype != torch.float32:
        raise L.VlsatError("scene_points must be float32 [Npts,3]")
    n, p = choice.shape
    pts = torch.empty(n, 3, p, dtype=torch.float32, device=scene_points.device)
    desc = torch.empty(n, 11, dtype=torch.float32, device=scene_points.device)
    L.check(lib.vlsat_prepare_objects(scene_points.data_ptr(), choice.data_ptr(), n, p, pts.data_ptr(), desc.data_ptr(),
                                      L.stream_ptr()))
    return pts, desc


def fc_edges(n_per_scene: Sequence[int], device) -> tuple:
    """-> edge_indices i64[2,E] (what Mmgnet.forward takes), batch_ids i64[N,1]."""
    lib = L.load()
    n = torch.tensor([0] + list(n_per_scene), dtype=torch.int64)
    node_ptr = torch.cumsum(n, 0)
    edge_ptr = torch.cumsum(n * (n - 1), 0)
    N, E, S = int(node_ptr[-1]), int(edge_ptr[-1]), len(n_per_scene)
    d_node, d_edge = node_ptr.to(torch.int32).to(device), edge_ptr.to(device)
    edges = torch.empty(2, E, dtype=torch.int64, device=device)
    bids = torch.empty(N, 1, dtype=torch.int64, device=device)
    L.check(lib.vlsat_fc_edges(d_node.data_ptr(), d_edge.data_ptr(), S, N, E, edges.data_ptr(), bids.data_ptr(),
                               L.stream_ptr()))
    return edges, bids
