#!/usr/bin/env python3
"""Throughput of the bench batch (64 x 40 x 256, L = 3, fp32) under every MODEL.* switch that changes the path's ops: which
configurations run on a slow path.   python tools/switch_scan.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vlsat_amd
from vlsat_amd import VLSATConfig, synth
from vlsat_amd.model import VLSATModel
for kw in ({}, {"feature_transform": True}, {"USE_RGB": True, "USE_NORMAL": True}, {"WITH_BN": True}, {"multi_rel_outputs": False}, {"GCN_AGGR": "mean"}, {"USE_GCN_EDGE": False}):
    cfg = VLSATConfig(N_LAYERS=3, **kw)
    m = VLSATModel(cfg, "cuda:0").load_state(synth.make_weights(cfg)).eval()
    scenes = [synth.make_scene(40, 256, 1000 + s) for s in range(64)]
    if cfg.dim_point > 3:
        for i, sc in enumerate(scenes):
            g = np.random.default_rng([i, cfg.dim_point])
            sc["obj_points"] = np.concatenate([sc["obj_points"], g.uniform(-1, 1, (40, cfg.dim_point - 3, 256)).astype(np.float32)], 1)
    b = synth.collate(scenes)
    d = {k: torch.from_numpy(v).to("cuda:0") for k, v in b.items()}
    try:
        for _ in range(2): m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(kw, f"{dt*1e3:.2f} ms/step, {64/dt:.0f} scenes/s", flush=True)
    except Exception as e:
        print(kw, "ERR", str(e)[:200])
    m.close()
