#!/usr/bin/env python3
"""One scene per call, end to end: the reference's validation() loop (batch_size = 1, src/model/model.py:185,201-211 ->
process_val, SGFN_MMG/model.py:458-480) on scenes of 9..80 objects, every scene a different graph -- forward + ranking +
counts per scene.  Compares
    workers = 0   the reference-compatible loop: numpy rank lists on the host after every scene (four host round trips)
    workers = K   evaluate.validation(workers=K): counts on the device, K scenes in flight on K streams / model replicas
and, for scale, the same scenes handed over as ONE batch.  Summaries must be identical in every mode.

    python tools/val_loop_probe.py [--scenes 120] [--gemm-precision fp32] [--workers 1,2,4,6,8]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402
from vlsat_amd import VLSATConfig, synth, evaluate as EV  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=120)
    ap.add_argument("--points", type=int, default=256)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--objects", default="9,80", help="range of objects per scene (3RScan: 9..80), or one number")
    ap.add_argument("--gemm-precision", default="fp32", choices=["fp32", "bf16x3", "bf16_mixed", "fp16_mixed", "bf16x3_attn1"])
    ap.add_argument("--workers", default="1,2,4,6,8")
    ap.add_argument("--merge", default="4,8,16", help="also: K workers each collating B consecutive one-scene batches into one call (evaluate.merge_batches)")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE", help="vlsat_debug_option of the model (replicas inherit it)")
    ap.add_argument("--no-hint", action="store_true", help="do not pass fc_sizes: the plan cache hashes the (device) edge list, one read-back per new graph")
    a = ap.parse_args()
    dev = "cuda:0"
    cfg = VLSATConfig(N_LAYERS=a.layers)
    model = VLSATModel(cfg, dev).load_state(synth.make_weights(cfg)).eval().set_gemm_precision(a.gemm_precision)
    for kv in a.debug_option:
        model.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    rng = np.random.default_rng(5)
    lo, hi = ([int(x) for x in a.objects.split(",")] * 2)[:2]
    sizes = rng.integers(lo, hi + 1, a.scenes)

    def item(scene_list, ns):
        b = synth.collate(scene_list)
        n, e = b["obj_points"].shape[0], b["edge_indices"].shape[1]
        g = np.random.default_rng([n, e, 7])
        it = {k: torch.from_numpy(v).to(dev) for k, v in b.items() if k != "edge_indices"}
        it.update(gt_class=torch.from_numpy(g.integers(0, 160, n)).to(dev), gt_rel_cls=torch.from_numpy((g.random((e, 26)) < 0.04).astype(np.int64)).to(dev),
                  edge_indices=torch.from_numpy(b["edge_indices"]).t().contiguous().to(dev))
        if not a.no_hint:
            it["fc_sizes"] = list(ns)
        else:
            it["n_scenes"] = len(ns)
        return it
    scenes = [synth.make_scene(int(n), a.points, seed=100 + i) for i, n in enumerate(sizes)]
    one_per_call = [item([s], [int(n)]) for s, n in zip(scenes, sizes)]
    # labels of the single batch = the per-scene labels concatenated, so that the summaries are comparable
    big = item(scenes, [int(n) for n in sizes])
    big["gt_class"] = torch.cat([it["gt_class"] for it in one_per_call])
    big["gt_rel_cls"] = torch.cat([it["gt_rel_cls"] for it in one_per_call])
    flops = sum(float(n) for n in sizes)

    def timed(fn, reps=3):
        fn()                                           # warm: plans of these graphs, allocator, kernel code objects
        torch.cuda.synchronize()
        best, out = 1e9, None
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best, out
    print(f"{a.scenes} scenes of {lo}..{hi} objects x {a.points} points, L={a.layers}, {a.gemm_precision}; mean {flops / a.scenes:.1f} objects; "
          f"{'edge list hashed (no fc_sizes hint)' if a.no_hint else 'fc_sizes hint'}; best of 3 passes, plans warm")
    t, ref = timed(lambda: EV.validation(model, one_per_call, device=dev, workers=0))
    print(f"  one scene per call, reference-compatible loop (host rank lists)   {a.scenes / t:8.1f} scenes/s   {t / a.scenes * 1e3:6.3f} ms/scene")
    for k in [int(x) for x in a.workers.split(",")]:
        t, got = timed(lambda: EV.validation(model, one_per_call, device=dev, workers=k))
        assert got == ref, "summaries differ: " + ", ".join(f"{q} {ref[q]!r} vs {got[q]!r}" for q in ref if got[q] != ref[q])
        print(f"  one scene per call, counts on the device, {k} in flight              {a.scenes / t:8.1f} scenes/s   {t / a.scenes * 1e3:6.3f} ms/scene")
    for bsz in [int(x) for x in a.merge.split(",") if x]:
        for k in (1, 2):
            t, got = timed(lambda: EV.validation(model, one_per_call, device=dev, workers=k, merge=bsz))
            worst = max(abs(got[q] - ref[q]) for q in ref)
            print(f"  one scene per item, {bsz} items collated per call, {k} in flight          {a.scenes / t:8.1f} scenes/s   {t / a.scenes * 1e3:6.3f} ms/scene"
                  f"   largest difference of a summary percentage: {worst:.4f}")
    t, got = timed(lambda: EV.validation(model, [big], device=dev, workers=1))
    worst = max(abs(got[k] - ref[k]) for k in ref)      # (batched and one-scene forwards differ in the last bits: a near-tie may move a rank)
    print(f"  all {a.scenes} scenes in ONE call (batched)                               {a.scenes / t:8.1f} scenes/s   {t / a.scenes * 1e3:6.3f} ms/scene"
          f"   largest difference of a summary percentage: {worst:.4f}")
    print("  " + ", ".join(f"{k} {ref[k]:.3f}" for k in ("obj_acc@1_3d", "rel_acc@1_3d", "tri_acc@50_3d", "mean_recall@100_3d", "tri_acc@50_2d")))


if __name__ == "__main__":
    main()
