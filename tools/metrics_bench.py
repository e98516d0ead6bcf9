#!/usr/bin/env python3
"""Time the GPU eval-ranking step on the cfg-2 batch (64 scenes x 40 objects, E = 99 840) with random
ground truth, next to the forward.   python tools/metrics_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import VLSATConfig, synth, metrics as M  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402

dev = "cuda:0"
cfg = VLSATConfig(N_LAYERS=3)
model = VLSATModel(cfg, dev).load_state(synth.make_weights(cfg)).eval()
b = synth.make_batch(64, 40, 256)
d = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
g = torch.Generator().manual_seed(0)
n, e = d["obj_points"].shape[0], d["edge_indices"].shape[1]
gt_cls = torch.randint(0, 160, (n,), generator=g).to(dev)
gt_rel = (torch.rand(e, 26, generator=g) < 0.05).long().to(dev)
edges = d["edge_indices"].t().contiguous()
out = model(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
for _ in range(2):
    M.eval_ranks(out[0], out[2], gt_cls, gt_rel, edges)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    r3 = M.eval_ranks(out[0], out[2], gt_cls, gt_rel, edges)
    r2 = M.eval_ranks(out[1], out[3], gt_cls, gt_rel, edges)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"ranking (3D + 2D) for 64 scenes: {dt * 1e3:.2f} ms  = {dt / 64 * 1e3:.3f} ms/scene "
      f"(reference CPU: ~116 s/scene at E=1560, SURVEY §6)")
s = M.summarize(r3["top_k_obj"].cpu(), r3["top_k_rel"].cpu(), r3["top_k_triplet"].cpu())
print({k: round(v, 2) for k, v in s.items()})
