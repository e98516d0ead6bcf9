#!/usr/bin/env python3
"""The object encoder alone (forward stopped after stage 1) at the bench batch: microseconds per launch in fp32 and in the
bf16 modes.  Under `rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python tools/pointnet_probe.py`
the counters of pointnet_kernel / pointnet_bf16_kernel give the LDS bank-conflict share (tools/pmc_raw.py prints them).
    python tools/pointnet_probe.py [--objects 2560] [--points 256] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import VLSATConfig, synth  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=2560)
    ap.add_argument("--points", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default="", help="another build of libvlsat_hip.so to load instead (A/B on one box)")
    a = ap.parse_args()
    if a.lib:
        from vlsat_amd import lib as L
        L.LIB_PATH = os.path.abspath(a.lib)
    dev = "cuda:0"
    cfg = VLSATConfig(N_LAYERS=1)
    per = 40
    scenes = [synth.make_scene(per, a.points, 10 + i) for i in range(a.objects // per)]
    d = {k: torch.from_numpy(v).to(dev) for k, v in synth.collate(scenes).items()}
    m = VLSATModel(cfg, dev).load_state(synth.make_weights(cfg)).eval()
    m.debug_stop_after(1)
    for mode in ("fp32", "bf16x3", "bf16_mixed"):
        m.set_gemm_precision(mode)
        for _ in range(3):
            m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.reps * 1e3
        fl = 213376.0 * d["obj_points"].shape[0] * a.points
        print(f"{mode:11s} {d['obj_points'].shape[0]} objects x {a.points} points: {us:8.1f} us per launch  {fl / us * 1e-6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
