#!/usr/bin/env python3
"""Where the single-rounding bf16 mode stops meeting 1e-2: formula weights with the GCN matrices scaled by s and LayerNorm
gains from U(0.3, 3) (synth.make_weights_stress), 64-scene bench batch, four scenes against the fp64 oracle.
    python tools/stress_scan.py [--scales 1,2,4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import VLSATConfig, synth  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402
from oracle import vlsat_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scales", default="1,1.5,2,3,4")
a = ap.parse_args()
cfg = VLSATConfig(N_LAYERS=3)
scenes = [synth.make_scene(40, 256, 1000 + s) for s in range(64)]
d = {k: torch.from_numpy(v).to("cuda:0") for k, v in synth.collate(scenes).items()}
N, E = 40, 1560
for sc in [float(x) for x in a.scales.split(",")]:
    w = synth.make_weights_stress(cfg, sc)
    w64 = O.to_torch(w, torch.float64)
    refs = {}
    for s in (0, 21, 42, 63):
        c = {k: torch.from_numpy(v) for k, v in synth.collate([scenes[s]]).items()}
        refs[s] = O.forward(w64, cfg, c["obj_points"].double(), c["obj_2d_feats"].double(), c["edge_indices"], c["descriptor"].double(), c["batch_ids"])
    m = VLSATModel(cfg, "cuda:0").load_state(w).eval()
    for mode, opts in (("fp32", {}), ("bf16x3", {}), ("bf16_mixed", {}), ("fp16_mixed", {}), ("bf16x3_attn1", {}), ("bf16_mixed", {"half_fmt": 0}), ("bf16", {})):
        m.set_gemm_precision(mode)
        try:
            for k, v in opts.items():
                m.debug_option(k, v)
        except Exception as ex:           # a lab switch (half_fmt) on the release library: that row needs `bench.py --lib`-style loading of
            print(f"scale {sc:4.1f} {mode:10s} {str(opts):16s} skipped: {str(ex)[-80:]}")      # tools/bin/libvlsat_hip_exp.so
            continue
        got = [o.cpu() for o in m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])]
        worst = [0.0] * 4
        for s, ref in refs.items():
            sl = [slice(s * N, (s + 1) * N)] * 2 + [slice(s * E, (s + 1) * E)] * 2
            for i in range(4):
                worst[i] = max(worst[i], float((got[i][sl[i]] - ref[i].float()).abs().max()))
        print(f"scale {sc:4.1f} {mode:10s} {str(opts):16s} obj3d {worst[0]:.2e} obj2d {worst[1]:.2e} rel3d {worst[2]:.2e} rel2d {worst[3]:.2e}", flush=True)
        for k in opts:
            m.debug_option(k, 1)
    m.close()
