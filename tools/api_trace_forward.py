#!/usr/bin/env python3
"""Workload for a rocprofv3 HIP-API trace of the call boundary (no hidden device-wide wait):
    rocprofv3 --hip-runtime-trace --stats -d OUT -o api -- python tools/api_trace_forward.py
40 scenes of different sizes, one per call, every call a graph the model has not seen (plan build + upload + forward),
then the same 40 again (plan cache hits), with NO synchronisation from this script between calls; the cache is kept at
8 plans so that plans are evicted and their workspaces recycled all the time.  The script itself synchronises exactly
SYNCS times (printed); every other hipDeviceSynchronize / hipStreamSynchronize / blocking hipMemcpy in the trace would
come from the library.  (edge lists are handed over on the HOST, like the reference's data loader yields them, so
the plan needs no D2H copy.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import VLSATConfig, synth  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402

dev = "cuda:0"
cfg = VLSATConfig(N_LAYERS=3)
model = VLSATModel(cfg, dev).load_state(synth.make_weights(cfg)).eval()
model.MAX_PLANS = 8
sizes = list(range(9, 49))
items = []
for i, n in enumerate(sizes):
    b = synth.collate([synth.make_scene(n, 128, 300 + i)])
    d = {k: torch.from_numpy(v).to(dev) for k, v in b.items() if k not in ("edge_indices", "batch_ids")}
    d["edge_indices"], d["batch_ids"] = torch.from_numpy(b["edge_indices"]), torch.from_numpy(b["batch_ids"])
    items.append(d)
syncs = 0
model(items[0]["obj_points"], items[0]["obj_2d_feats"], items[0]["edge_indices"], items[0]["descriptor"], items[0]["batch_ids"])
torch.cuda.synchronize(); syncs += 1
model._drop_plans()
outs = []
for rep in range(2):
    for it in items:
        outs.append(model(it["obj_points"], it["obj_2d_feats"], it["edge_indices"], it["descriptor"], it["batch_ids"]))
torch.cuda.synchronize(); syncs += 1
print(f"SYNCS {syncs}  forwards {len(outs)}  plan stats {model.plan_stats}")
