#!/usr/bin/env python3
"""Soak of the batched forward: the bench batch N times per precision mode (both streams, as deployed), every output compared
bit for bit with the first run's -- a race between the two streams, in the split-K exchange or in a counted wait would show as
a mismatch; a hang as a timeout.    python tools/soak_forward.py [--iters 300]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import VLSATConfig, synth  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--scenes", type=int, default=64)
    a = ap.parse_args()
    cfg = VLSATConfig(N_LAYERS=3)
    w = synth.make_weights(cfg)
    batch = synth.collate([synth.make_scene(40, 256, 1000 + s) for s in range(a.scenes)])
    small = [synth.collate([synth.make_scene(n, 128, 3000 + n)]) for n in (9, 23, 40, 57, 80)]
    bad_total = 0
    for mode in ("fp32", "bf16x3", "bf16_mixed", "fp16_mixed"):
        m = VLSATModel(cfg, "cuda:0").load_state(w).eval().set_gemm_precision(mode)
        d = {k: torch.from_numpy(v).to("cuda:0") for k, v in batch.items()}
        ds = [{k: torch.from_numpy(v).to("cuda:0") for k, v in b.items()} for b in small]
        call = lambda x: m(x["obj_points"], x["obj_2d_feats"], x["edge_indices"], x["descriptor"], x["batch_ids"])
        ref = [o.clone() for o in call(d)]
        refs = [[o.clone() for o in call(x)] for x in ds]
        torch.cuda.synchronize()
        bad, t0 = 0, time.perf_counter()
        for i in range(a.iters):
            out = call(d)
            bad += sum(int(not torch.equal(o, r)) for o, r in zip(out, ref))
            x = i % len(ds)                                   # a one-scene call between the big ones (plan switch, small kernels)
            out = call(ds[x])
            bad += sum(int(not torch.equal(o, r)) for o, r in zip(out, refs[x]))
        torch.cuda.synchronize()
        print(f"{mode}: {a.iters} batched + {a.iters} one-scene forwards in {time.perf_counter() - t0:.1f} s, outputs differing from the first run: {bad}", flush=True)
        bad_total += bad
        m.close()
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
