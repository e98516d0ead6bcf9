#!/usr/bin/env python3
"""Soak of the opt-in hipGraph entry point (vlsat_forward_graph / VLSATModel.forward_replay): capture and replay many
graphs of different sizes back to back while the plan cache evicts, the pattern tools/latency_probe.py uses.

    python tools/graph_replay_soak.py [--rounds 4]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import VLSATConfig, synth  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--scenes", type=int, default=60)
    a = ap.parse_args()
    dev = "cuda:0"
    cfg = VLSATConfig(N_LAYERS=3)
    model = VLSATModel(cfg, dev).load_state(synth.make_weights(cfg)).eval()
    rng = np.random.default_rng(5)
    sizes = rng.integers(9, 81, a.scenes)
    items = [{k: torch.from_numpy(v).to(dev) for k, v in synth.collate([synth.make_scene(int(n), 256, seed=100 + i)]).items()}
             for i, n in enumerate(sizes)]
    ref = [[o.clone() for o in model(it["obj_points"], it["obj_2d_feats"], it["edge_indices"], it["descriptor"], it["batch_ids"])] for it in items]
    torch.cuda.synchronize()
    bad = 0
    for r in range(a.rounds):
        model._drop_plans()
        for rep in range(2):                      # first pass captures, second replays
            for it, want in zip(items, ref):
                got = model.forward_replay(it["obj_points"], it["obj_2d_feats"], it["edge_indices"], it["descriptor"], it["batch_ids"])
                bad += sum(int(not torch.equal(g, w)) for g, w in zip(got, want))
        torch.cuda.synchronize()
        print(f"round {r}: {2 * len(items)} graph launches, mismatching outputs so far: {bad}", flush=True)
    print("plan cache:", model.plan_stats)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
