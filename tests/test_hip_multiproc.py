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


def _bench(n_ranks, scenes_per_rank, port):
    env = dict(os.environ, VLSAT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tail = ["bench.py", "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--no-cpu", "--scenes", str(scenes_per_rank)]
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
