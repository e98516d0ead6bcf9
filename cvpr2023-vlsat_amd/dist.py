"""Scene sharding across the GPUs of one node.

Scenes are independent units at eval (reference ``validation()`` runs ``batch_size=1``,
``src/model/model.py:185``), so the path shards with NO data-path collective:
every rank owns a contiguous slice of the scene list, runs the forward on its own MI355X and
only the final fixed-length metrics vector is summed with one all-reduce (RCCL over xGMI on
GPUs, gloo in the CPU tests).  One process per GPU, launched by ``torch.distributed.run``.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist

METRIC_FIELDS = ("scenes", "nodes", "edges", "sum_obj3d", "sum_obj2d", "sum_rel3d", "sum_rel2d",
                 "top1_agree_obj", "top1_agree_rel")


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str | None = None) -> Tuple[int, int, int]:
    rank, local, world = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:                                               # "nccl" is RCCL on ROCm
            backend = os.environ.get("VLSAT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced split of ``n_items`` scenes: ranks < n_items % world get one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def scene_metrics(outputs, n_scenes: int) -> torch.Tensor:
    """Fixed-length fp64 vector for one rank's batch (device of the outputs).  Sums are additive
    over scenes, so the all-reduced vector is independent of how scenes were sharded:
    counts, output checksums, and 3D-vs-2D top-1 agreement counts (the quantities
    ``validation()`` accumulates are per-scene additive in the same way, reference
    src/model/model.py:214-242)."""
    obj3, obj2, rel3, rel2 = outputs
    dev = obj3.device
    if obj3.is_cuda:                      # the library's two-launch reduction (vlsat_scene_checksums); below: its CPU twin
        return _scene_metrics_hip(obj3, obj2, rel3, rel2, n_scenes)
    v = torch.zeros(len(METRIC_FIELDS), dtype=torch.float64, device=dev)
    v[0] = n_scenes
    v[1] = obj3.shape[0]
    v[2] = rel3.shape[0]
    v[3] = obj3.double().sum()
    v[4] = obj2.double().sum()
    v[5] = rel3.double().sum()
    v[6] = rel2.double().sum()
    v[7] = (obj3.argmax(-1) == obj2.argmax(-1)).sum()
    if rel3.shape[0]:
        v[8] = (rel3.argmax(-1) == rel2.argmax(-1)).sum()
    return v


_CHECKSUM_SCRATCH = {}


def _scene_metrics_hip(obj3, obj2, rel3, rel2, n_scenes: int) -> torch.Tensor:
    from . import lib as L
    lib = L.load()
    dev = obj3.device
    for t in (obj3, obj2, rel3, rel2):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise L.VlsatError("scene_metrics: outputs must be contiguous fp32 tensors on one GPU")
    scratch = _CHECKSUM_SCRATCH.get(dev)
    if scratch is None:
        scratch = _CHECKSUM_SCRATCH[dev] = torch.empty(256 * 6, dtype=torch.float64, device=dev)
    v = torch.empty(len(METRIC_FIELDS), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.vlsat_scene_checksums(obj3.data_ptr(), obj2.data_ptr(), obj3.shape[0], obj3.shape[1],
                                          rel3.data_ptr() if rel3.shape[0] else None, rel2.data_ptr() if rel2.shape[0] else None,
                                          rel3.shape[0], max(rel3.shape[1], 1), n_scenes, v.data_ptr(), scratch.data_ptr(),
                                          L.stream_ptr()))
    return v


class NativeComm:
    """The library's own RCCL communicator (``vlsat_comm_*`` / ``vlsat_metrics_allreduce``, include/vlsat.h): what a host
    without PyTorch would use for the one collective of the path.  Bootstrap: rank 0 makes the 128-byte unique id and
    the ranks exchange it -- here through the already initialised ``torch.distributed`` group (any backend), a file or
    a pipe does as well.  One communicator per process, on the current GPU."""

    def __init__(self, rank: int, world: int):
        import ctypes as C
        from . import lib as L
        self._L, self._lib, self.world = L, L.load(), world
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            L.check(self._lib.vlsat_comm_unique_id(C.c_void_p(ident.data_ptr())))
        if world > 1:
            if not dist.is_initialized():
                raise L.VlsatError("NativeComm: the ranks need a torch.distributed group (or another channel) to share the id")
            if dist.get_backend() == "gloo":
                dist.broadcast(ident, src=0)
            else:
                d = ident.cuda()
                dist.broadcast(d, src=0)
                ident = d.cpu()
        self._comm = C.c_void_p()
        L.check(self._lib.vlsat_comm_init(C.c_void_p(ident.data_ptr()), world, rank, C.byref(self._comm)))

    def allreduce(self, v: torch.Tensor) -> torch.Tensor:
        if not (v.is_cuda and v.dtype == torch.float64 and v.is_contiguous()):
            raise self._L.VlsatError("NativeComm.allreduce: contiguous fp64 CUDA tensor expected")
        self._L.check(self._lib.vlsat_metrics_allreduce(self._comm, v.data_ptr(), v.numel(), self._L.stream_ptr()))
        return v

    def close(self):
        if self._comm:
            self._lib.vlsat_comm_destroy(self._comm)
            self._comm = None


_native: "NativeComm | None" = None


def use_native_allreduce(rank: int, world: int) -> None:
    """Route allreduce_metrics through the library's RCCL entry point (bench.py --native-allreduce)."""
    global _native
    _native = NativeComm(rank, world)


def allreduce_metrics(v: torch.Tensor) -> torch.Tensor:
    """The single collective of the path: sum of the metrics vector over ranks."""
    if _native is not None and v.is_cuda:
        return _native.allreduce(v)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if v.is_cuda and dist.get_backend() == "gloo":      # test rigs without RCCL: reduce on the host
            h = v.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            v.copy_(h)
        else:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return v


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return x


def minmax_over_ranks(x: float, device):
    """(min, max) of a per-rank scalar: bench.py reports the spread of the ranks' step times."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dv = "cpu" if dist.get_backend() == "gloo" else device
        lo, hi = torch.tensor([x], dtype=torch.float64, device=dv), torch.tensor([x], dtype=torch.float64, device=dv)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return float(lo.item()), float(hi.item())
    return x, x
