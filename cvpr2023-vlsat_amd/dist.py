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


def allreduce_metrics(v: torch.Tensor) -> torch.Tensor:
    """The single collective of the path: sum of the metrics vector over ranks."""
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
