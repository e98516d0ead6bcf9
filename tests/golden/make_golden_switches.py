#!/usr/bin/env python3
"""Golden vectors for the config switches that change the path's ops (SURVEY.md 8a switch table), made by running
the REAL reference (/root/reference, imported read-only through the stand-ins of make_golden.py) on CPU here:

    python tests/golden/make_golden_switches.py        ->  tests/golden/switch_*.npz

One small ragged batch (4 + 6 objects x 32 points, per-scene reference calls concatenated = validation()'s
batch_size=1 contract) per switch; weights and inputs are the seeded formulas of vlsat_amd.synth, so only
outputs are stored.
  switch_with_bn        MODEL.WITH_BN=true          BatchNorm1d(eval) after fc1/fc2 of both relation heads
  switch_no_gcn_edge    MODEL.USE_GCN_EDGE=false    gate MLP on the projected query alone (64->128->32)
  switch_single_rel     MODEL.multi_rel_outputs=false, 27 relation classes: log_softmax head
  switch_rgb_normal     MODEL.USE_RGB=USE_NORMAL=true: 9 point channels
  switch_feature_transform  MODEL.feature_transform=true: STNkd 64x64 transform after conv1 of all three encoders
  switch_all            all of the above together, GCN_AGGR=mean, L=1
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from vlsat_amd import VLSATConfig, synth  # noqa: E402

def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MG.install_standins()
    for name, kw in synth.SWITCH_CASES.items():
        cfg = VLSATConfig(**kw)
        over = {k: v for k, v in kw.items() if k in ("WITH_BN", "USE_GCN_EDGE", "multi_rel_outputs", "USE_RGB", "USE_NORMAL", "feature_transform")}
        m = MG.build_reference(cfg.N_LAYERS, cfg.GCN_AGGR, num_rel=cfg.num_rel_class, **over)
        MG.load_formula_weights(m, cfg)
        per = [MG.run(m, synth.collate([s])) for s in synth.switch_scenes(cfg)]
        cat = {k: np.concatenate([p[k] for p in per], 0) for k in ("obj3d", "obj2d", "rel3d", "rel2d")}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **cat)
        print(name, {k: v.shape for k, v in cat.items()})


if __name__ == "__main__":
    main()
