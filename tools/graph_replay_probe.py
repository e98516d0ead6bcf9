#!/usr/bin/env python3
"""Does replaying a captured forward (vlsat_forward_graph / VLSATModel.forward_replay) beat launching it?  One scene, the
same input buffers, timed call by call with a host sync after each: six plain forwards, then eight replays (the first of
which captures).    python tools/graph_replay_probe.py      -> profiles/r02_probes/graph_replay.txt"""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd
from vlsat_amd import VLSATConfig, synth
from vlsat_amd.model import VLSATModel
cfg = VLSATConfig(N_LAYERS=3)
m = VLSATModel(cfg, "cuda:0").load_state(synth.make_weights(cfg)).eval()
for n in (12, 40):
    b = synth.collate([synth.make_scene(n, 256, 1)])
    d = {k: torch.from_numpy(v).to("cuda:0") for k, v in b.items()}
    args = (d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
    for _ in range(3): m(*args)
    torch.cuda.synchronize()
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); m(*args); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    tg = []
    for i in range(8):
        t0 = time.perf_counter(); m.forward_replay(*args); torch.cuda.synchronize(); tg.append(time.perf_counter() - t0)
    print(n, "objects: forward", [f"{1e3*t:.2f}" for t in ts], " replay (first = capture)", [f"{1e3*t:.2f}" for t in tg])
