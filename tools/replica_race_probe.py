#!/usr/bin/env python3
"""Are K model replicas driven from K host threads / streams bit-identical to the single-threaded forward?
Every scene runs once on the main model (reference logits), then every worker thread runs ALL scenes on its replica
concurrently with the others, `--passes` times; any output that is not bit-identical is reported with the scene, the
worker, the output and the size of the difference.

    python tools/replica_race_probe.py [--scenes 200] [--workers 5] [--passes 4] [--gemm-precision fp32]"""
import argparse
import ctypes as C
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsat_amd  # noqa: E402,F401
from vlsat_amd import VLSATConfig, synth, lib as VL  # noqa: E402
from vlsat_amd.model import VLSATModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=200)
    ap.add_argument("--workers", type=int, default=5)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--points", type=int, default=256)
    ap.add_argument("--gemm-precision", default="fp32")
    ap.add_argument("--debug-option", action="append", default=[])
    ap.add_argument("--no-hint", action="store_true")
    ap.add_argument("--fingerprint", action="store_true", help="also fingerprint the plan's workspace buffers after every forward (which stage went wrong)")
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--max-plans", type=int, default=0, help="plan cache capacity of every replica (default: the model's 64)")
    a = ap.parse_args()
    dev = "cuda:0"
    cfg = VLSATConfig(N_LAYERS=a.layers)
    model = VLSATModel(cfg, dev).load_state(synth.make_weights(cfg)).eval().set_gemm_precision(a.gemm_precision)
    rng = np.random.default_rng(5)
    sizes = rng.integers(9, 81, a.scenes)
    items = []
    for i, n in enumerate(sizes):
        b = synth.collate([synth.make_scene(int(n), a.points, seed=100 + i)])
        it = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
        it["fc"] = None if a.no_hint else [int(n)]
        items.append(it)
    models = [model] + model.replicas(a.workers - 1)
    for m in models:
        if a.max_plans:
            m.MAX_PLANS = a.max_plans
        for o in a.debug_option:
            k, v = o.split("=")
            m.debug_option(k, int(v))

    NAMES = ["F", "H1", "X3", "X2", "E3", "E2", "KP", "NP", "G", "Hbig", "On", "Oe", "Qe", "KVe"]

    class Alias:                                             # a plan buffer seen as a tensor (no copy)
        def __init__(self, ptr, rows, cols, ld):
            self.__cuda_array_interface__ = {"shape": (rows, cols), "strides": (ld * 4, 4), "typestr": "<i4", "data": (ptr, False), "version": 2}

    alias = {}

    def views(k, i):
        if (k, i) not in alias:
            m, it = models[k], items[i]
            n = it["obj_points"].shape[0]
            plan = m._plan(it["edge_indices"], it["batch_ids"], n, a.points, it["fc"])
            v = []
            for nm in NAMES:
                ptr, rows, cols, ld = C.c_void_p(), C.c_int64(), C.c_int32(), C.c_int32()
                VL.check(m._lib.vlsat_debug_buffer(plan.handle, nm.encode(), C.byref(ptr), C.byref(rows), C.byref(cols), C.byref(ld)))
                v.append(torch.as_tensor(Alias(ptr.value, rows.value, cols.value, ld.value), device=dev) if rows.value and ptr.value else None)
            alias[(k, i)] = v
        return alias[(k, i)]

    def prints(k, i, out_row):                               # exact (integer) fingerprints, on the current stream
        for b, v in enumerate(views(k, i)):
            if v is not None:
                out_row[b] = v.sum(dtype=torch.int64)

    def fwd(m, it):
        return m(it["obj_points"], it["obj_2d_feats"], it["edge_indices"], it["descriptor"], it["batch_ids"], istrain=False, fc_sizes=it["fc"])
    with torch.no_grad():
        ref = [tuple(t.clone() for t in fwd(model, it)) for it in items]
        ref_fp = torch.zeros(a.scenes, len(NAMES), dtype=torch.int64, device=dev)
        ref_F = [None] * a.scenes
        if a.fingerprint:
            for i, it in enumerate(items):
                fwd(model, it)
                prints(0, i, ref_fp[i])
                ref_F[i] = views(0, i)[0].clone()
        again = [fwd(model, it) for it in items]
        torch.cuda.synchronize()
        rerun = sum(not all(torch.equal(x, y) for x, y in zip(r, g)) for r, g in zip(ref, again))
        print(f"single thread, same handle, second pass: {rerun} of {a.scenes} scenes differ")
    bad, lock = [], threading.Lock()

    def work(k, p):
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s), torch.no_grad():
            order = np.random.default_rng([k, p]).permutation(a.scenes)
            diff = torch.zeros(a.scenes, 4, device=dev)            # filled on the device: no host wait between forwards
            fp = torch.zeros(a.scenes, len(NAMES), dtype=torch.int64, device=dev)
            fp2 = torch.zeros(a.scenes, len(NAMES), dtype=torch.int64, device=dev)
            snapF = [None] * a.scenes
            for i in order:
                out = fwd(models[k], items[i])
                for j, (x, y) in enumerate(zip(ref[i], out)):
                    diff[i, j] = (x - y).abs().max()
                if a.fingerprint:
                    prints(k, i, fp[i])
                    prints(k, i, fp2[i])                      # read again: a transient (stale) read, or has memory changed?
                    snapF[i] = views(k, i)[0].clone()
            s.synchronize()
            if a.fingerprint:
                for i in torch.nonzero((fp != ref_fp).any(1) | (fp2 != ref_fp).any(1)).view(-1).tolist():
                    with lock:
                        print(f"  pass {p} worker {k} scene {i} ({sizes[i]} objects): buffers that differ: "
                              + " ".join(nm for nm, x, y in zip(NAMES, fp[i].tolist(), ref_fp[i].tolist()) if x != y) + " | second read: "
                              + " ".join(nm for nm, x, y in zip(NAMES, fp2[i].tolist(), ref_fp[i].tolist()) if x != y), flush=True)
                        d = torch.nonzero(snapF[i] != ref_F[i])
                        if len(d):
                            r, c = d[:, 0], d[:, 1]
                            got, want = snapF[i][r, c].view(torch.float32), ref_F[i][r, c].view(torch.float32)
                            print(f"      F copy: {len(d)} words differ, rows {r.min().item()}..{r.max().item()} ({len(r.unique())} rows), cols {c.min().item()}..{c.max().item()}; "
                                  f"got[:6] {got[:6].tolist()} want[:6] {want[:6].tolist()}; zeros in got {(got == 0).sum().item()}, got<want {(got < want).sum().item()}", flush=True)
            for i, j in torch.nonzero(diff).tolist():
                with lock:
                    bad.append((p, k, int(i), int(sizes[i]), j, diff[i, j].item()))
    for p in range(a.passes):
        ts = [threading.Thread(target=work, args=(k, p)) for k in range(a.workers)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        print(f"pass {p}: {len(bad)} mismatching outputs so far; plan builds {[m.plan_stats['builds'] for m in models]}")
    for b in bad[:40]:
        print("  pass %d worker %d scene %d (%d objects) output %d  max |diff| %.3e" % b)


if __name__ == "__main__":
    main()
