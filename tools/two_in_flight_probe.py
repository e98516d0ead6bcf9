"""Steps of the 64-scene batch with K handles (model + replicas) on K caller streams, step i on handle i % K: the tail of one step (where
only the 2D edge lane still has work) runs under the head of the next.  Prints scenes/s for K = 1, 2, 3 in interleaved repetitions.
    python tools/two_in_flight_probe.py --gemm-precision bf16_mixed [--steps 40] [--reps 3]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import synth  # noqa: E402
from vlsat_amd.config import VLSATConfig  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gemm-precision", default="bf16_mixed")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--scenes", type=int, default=64)
    ap.add_argument("--max-k", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = VLSATConfig(N_LAYERS=3)
    model = VLSATModel(cfg, str(dev)).load_state(synth.make_weights(cfg)).eval()
    model.set_gemm_precision(a.gemm_precision)
    models = [model] + model.replicas(a.max_k - 1)
    batch = synth.collate([synth.make_scene(40, 256, 1000 + s) for s in range(a.scenes)])
    d = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    streams = [torch.cuda.Stream(device=dev) for _ in range(a.max_k)]

    def run(k, steps):
        outs = [None] * k
        for i in range(steps):
            j = i % k
            with torch.cuda.stream(streams[j]):
                outs[j] = models[j](d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
        return outs

    ref = None
    for k in range(1, a.max_k + 1):
        outs = run(k, 2 * k)
        torch.cuda.synchronize()
        for o in outs:
            if ref is None:
                ref = [t.clone() for t in o]
            assert all(torch.equal(x, y) for x, y in zip(o, ref)), "replica output differs"
    print(f"{a.gemm_precision}, {a.scenes} scenes x 40 objects x 256 points, {a.steps} steps per measurement; outputs of every handle bit-identical")
    for rep in range(a.reps):
        for k in range(1, a.max_k + 1):
            run(k, 4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(k, a.steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"  rep {rep}  {k} in flight   {a.scenes * a.steps / dt:9.1f} scenes/s   {dt / a.steps * 1e3:7.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
