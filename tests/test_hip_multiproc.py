"""The multi-process path THROUGH THE HIP LIBRARY on one GPU: what `bench.py --gpus N` does on an 8-GPU node
(one process per GPU, scenes sharded, one all-reduce of the additive metrics vector -- the sums `validation()`
derives from the concatenated per-scene results, reference src/model/model.py:214-242), here as two ranks that share
GPU 0 and reduce over gloo (RCCL needs one device per rank).  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(n_ranks, scenes_per_rank, port, backend="gloo", extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend:
        env["VLSAT_DIST_BACKEND"] = backend
    else:
        env.pop("VLSAT_DIST_BACKEND", None)
    tail = ["bench.py", "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--no-cpu", "--scenes", str(scenes_per_rank)] + list(extra)
    if n_ranks == 1:
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + tail
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_reproduce_the_single_process_metrics():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    two = _bench(2, 8, 29631)         # ranks own scenes 0..7 and 8..15
    one = _bench(1, 16, 0)            # the same 16 scenes in one process
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["scenes_per_gpu"] == 8
    assert two["value"] > 0 and two["steps"] == 2
    a, b = two["metrics_allreduced"], one["metrics_allreduced"]
    assert a["scenes"] == 16 and b["scenes"] == 16 and a["nodes"] == b["nodes"] == 16 * 40 and a["edges"] == b["edges"]
    for k in ("sum_obj3d", "sum_obj2d", "sum_rel3d", "sum_rel2d"):
        assert abs(a[k] - b[k]) <= 1e-3 * max(1.0, abs(b[k])), (k, a[k], b[k])
    for k in ("top1_agree_obj", "top1_agree_rel"):
        assert abs(a[k] - b[k]) <= 2, (k, a[k], b[k])


def test_eight_ranks_dry_run_of_the_drivers_scaling_command():
    """BASELINE configs[3] as the driver launches it -- `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`, 64 scenes
    per rank = 512 scenes -- with the eight ranks sharing the one GPU of the test box and gloo for the one all-reduce: shard
    sizes, the 512-scene total and the all-reduced counts are those of the real run, so the first 8-GPU run cannot fail on
    host logic (RCCL itself needs one device per rank: test_two_gpus_over_rccl_native_allreduce).  The checksums of an
    eight-rank run are then compared with ONE process over the same scenes (8 x 8 scenes against 64)."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    eight = _bench(8, 64, 29651, extra=["--no-extra"])
    assert eight["n_gpus"] == 8 and eight["scaling"] == "weak" and eight["config"]["scenes_per_gpu"] == 64
    assert eight["config"]["parallelism"] == "scene-sharded x8" and eight["value"] > 0 and eight["steps"] == 2
    a = eight["metrics_allreduced"]
    assert a["scenes"] == 512 and a["nodes"] == 512 * 40 and a["edges"] == 512 * 1560
    lo, hi = eight["rank_ms_per_step_min_max"]
    assert 0 < lo <= hi
    small = _bench(8, 8, 29661, extra=["--no-extra"])      # ranks own scenes 0..7, 8..15, ... 56..63
    one = _bench(1, 64, 0, extra=["--no-extra"])           # the same 64 scenes in one process
    a, b = small["metrics_allreduced"], one["metrics_allreduced"]
    assert a["scenes"] == b["scenes"] == 64 and a["nodes"] == b["nodes"] and a["edges"] == b["edges"]
    for k in ("sum_obj3d", "sum_obj2d", "sum_rel3d", "sum_rel2d"):
        assert abs(a[k] - b[k]) <= 1e-3 * max(1.0, abs(b[k])), (k, a[k], b[k])
    for k in ("top1_agree_obj", "top1_agree_rel"):
        assert abs(a[k] - b[k]) <= 4, (k, a[k], b[k])


def test_two_gpus_over_rccl_native_allreduce():
    """The run the driver launches on a multi-GPU node, at N = 2: one rank per GPU, RCCL for torch.distributed AND for the
    library's own collective (bench.py --native-allreduce -> vlsat_metrics_allreduce on a 2-rank communicator).  Needs two
    GPUs: on the one-GPU test box this SKIPS -- loudly, because it means RCCL has still only ever seen one rank here
    (tests/test_hip_round2.py::test_native_rccl_allreduce_single_rank)."""
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the HIP path cannot run and there is no fallback")
    if torch.cuda.device_count() < 2:
        pytest.skip(f"ONLY {torch.cuda.device_count()} GPU VISIBLE: the 2-rank RCCL all-reduce over xGMI was NOT exercised")
    two = _bench(2, 8, 29641, backend="", extra=["--native-allreduce"])
    one = _bench(1, 16, 0)
    assert two["n_gpus"] == 2 and two["allreduce"].startswith("vlsat_metrics_allreduce")
    lo, hi = two["rank_ms_per_step_min_max"]
    assert 0 < lo <= hi
    a, b = two["metrics_allreduced"], one["metrics_allreduced"]
    assert a["scenes"] == 16 and a["nodes"] == b["nodes"] and a["edges"] == b["edges"]
    for k in ("sum_obj3d", "sum_obj2d", "sum_rel3d", "sum_rel2d"):
        assert abs(a[k] - b[k]) <= 1e-3 * max(1.0, abs(b[k])), (k, a[k], b[k])
    assert two["evaluation"]["metrics"]["scenes"] == 16
