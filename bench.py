#!/usr/bin/env python3
"""bench.py -- scenes/s of the VL-SAT eval forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (Mmgnet.forward eval, all four outputs) over one batch of
synthetic 3RScan-shaped scenes per GPU: BASELINE.json configs[1] = 64 scenes x 40 objects x 256
points, fully-connected edges (E = 1560 per scene), 3 GNN layers, fp32 (exact-fp32 MFMA).
Inputs and weights are resident in HBM before the timed region.  Multi-GPU: scenes are sharded
(weak scaling, 64 scenes per GPU, no data-path collective); the metrics vector is all-reduced
once per step over RCCL.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import vlsat_amd  # noqa: E402
from vlsat_amd import VLSATConfig, synth, dist as vdist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense (never the 2:1-sparse figure)
# what the bf16 matrix pipe SUSTAINS on these boxes with registers only and random operands (tools/mfma_peak_bf16.hip,
# profiles/r04_probes/mfma_peak_bf16.txt: 1.78-1.83 PF; the clock drops to ~1.8 GHz under matrix load) -- printed next to the
# 2.5 PF headline peak as `peak_sustained`, never instead of it
SUSTAINED_BF16_MFMA_TFLOPS = 1800.0
# the fp32 matrix pipe at the clock the chip holds under these GEMMs with real operands (2.04-2.09 GHz instead of 2.4: in-kernel
# s_memtime against the wall clock, tools/gemm_clock_probe.py, profiles/r05_probes/gemm_clock_probe.txt) -- same role
SUSTAINED_FP32_MFMA_TFLOPS = 136.0
# peak of the mode's matrix work counted in ALGORITHMIC flops: split-bf16 issues three bf16 MFMAs per product
MODE_PEAK = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3, "bf16": PEAK_BF16_MFMA_TFLOPS,
             "bf16_mixed": PEAK_BF16_MFMA_TFLOPS, "bf16x3_attn1": PEAK_BF16_MFMA_TFLOPS / 3, "fp16_mixed": PEAK_BF16_MFMA_TFLOPS}
MODE_DTYPE = {"fp32": "f32",
              "bf16x3": "bf16x3 (split-bf16 MFMA operands, 3 MFMAs per product, f32 accumulate; softmax/LN and HBM tensors f32)",
              "bf16": "bf16 (single-rounded bf16 MFMA operands, f32 accumulate; softmax/LN and HBM tensors f32)",
              "bf16_mixed": "bf16 on edge-row matrix work + bf16x3 on node rows (f32 accumulate; softmax/LN and HBM tensors f32)",
              "bf16x3_attn1": "bf16x3 everywhere except a single-rounded bf16 edge cross-attention (f32 accumulate; softmax/LN f32)",
              "fp16_mixed": "f16 on edge-row matrix work (v_mfma_f32_32x32x16_f16: the bf16 rate) + bf16x3 on node rows (f32 accumulate; softmax/LN f32)"}


def f_alg(n, p, e, l):
    """Algorithmic (minimal-algebra, eval-live) FLOPs per scene: SURVEY.md §8a/§8d F_alg."""
    return (213376 * n * p + 774144 * n + 297728 * e + 524288 * n + 2816 * n * n
            + l * (2 * (2097152 * n + 2048 * n * n) + 2 * (2949120 * e + 4849664 * n) + (2097152 * e + 2048 * e * e))
            + 2 * 799744 * e + 2 * 163840 * n)


def oracle_all_scenes(cfg, batch_np, threads):
    """fp32 CPU oracle on EVERY scene of this rank's batch (the parity figure of the JSON line covers all of them)."""
    from oracle import vlsat_oracle as O
    torch.set_num_threads(threads)
    c = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    return O.forward(O.to_torch(synth.make_weights(cfg)), cfg, c["obj_points"], c["obj_2d_feats"], c["edge_indices"],
                     c["descriptor"], c["batch_ids"])


def cpu_baseline(cfg, n_obj, n_pts, budget_s=12.0, max_scenes=96, sweep_s=5.0):
    """The CPU oracle (torch fp32 port of the reference, one scene per call like validation())
    timed on this box's host cores on a bounded sample of the same workload.  torch's intra-op
    pool is tried at a few sizes (one scene each) and the fastest is used for the timed sample:
    with all cores of a many-core host the small per-scene ops are dominated by thread
    synchronisation, which would understate what the CPU can do.  The sample is ~12 s of CPU work
    (as many scenes as fit, at most 96), not a fixed handful: the hosts are shared and a six-scene
    sample moved by +-30 % between runs."""
    from oracle import vlsat_oracle as O
    w = O.to_torch(synth.make_weights(cfg))
    ncpu = os.cpu_count() or 1

    def run(seed):
        b = {k: torch.from_numpy(v) for k, v in synth.make_batch(1, n_obj, n_pts, seed0=seed).items()}
        t0 = time.perf_counter()
        out = O.forward(w, cfg, b["obj_points"], b["obj_2d_feats"], b["edge_indices"], b["descriptor"], b["batch_ids"])
        return time.perf_counter() - t0, out

    torch.set_num_threads(min(8, ncpu))
    _, first = run(1000)                                  # warm-up + the parity reference for scene 0
    trials = {}
    for t in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu)}):     # (the sweep itself is capped at sweep_s seconds)
        torch.set_num_threads(t)
        trials[t] = run(1001)[0]
        if sum(trials.values()) > sweep_s:
            break
    best = min(trials, key=trials.get)
    torch.set_num_threads(best)
    t_used, n_done = 0.0, 0
    for s in range(max_scenes):
        t_used += run(1002 + s)[0]
        n_done += 1
        if t_used > budget_s:
            break
    return {"value": n_done / t_used, "unit": "scenes/s", "cores": best, "kind": "port",
            "sample": f"{n_done} scenes of {n_obj} objects x {n_pts} points (L={cfg.N_LAYERS}), one scene per call; "
                      f"thread sweep s/scene {{{', '.join(f'{k}: {v:.2f}' for k, v in trials.items())}}} on a "
                      f"{ncpu}-cpu host; torch {torch.__version__} CPU fp32"}, first


PMC_TAG = {("cfg2", "fp32"): "_bench_pmc.json", ("cfg2", "bf16x3"): "_cfg3_bf16x3_pmc.json", ("cfg2", "bf16_mixed"): "_cfg3_bf16_mixed_pmc.json", ("cfg2", "bf16x3_attn1"): "_cfg3_bf16x3_attn1_pmc.json", ("cfg2", "fp16_mixed"): "_cfg3_fp16_mixed_pmc.json",
           ("cfg5", "fp32"): "_cfg5_fp32_pmc.json", ("cfg5", "bf16_mixed"): "_cfg5_bf16_mixed_pmc.json"}


def committed_traffic(workload, mode, dom):
    """HBM bytes per launch of kernel class `dom` from the newest committed rocprofv3 PMC summary of this workload and mode
    (profiles/rNN_*_pmc.json, written by tools/profile_run.sh from FETCH_SIZE / WRITE_SIZE passes of this very command:
    FETCH_SIZE x 2 per MI355X_MICROARCH.md + WRITE_SIZE).  NOT measured in this run: the caller labels it so, and `stale` says
    whether the summary was collected on another build than the one running now (the summary's `collected_on` stamp against
    vlsat_amd.lib.identity(): source digest first, library digest second; True when the summary has no stamp)."""
    tag = PMC_TAG.get((workload, mode))
    d = os.path.join(ROOT, "profiles")
    if not tag or not os.path.isdir(d):
        return None, None, None
    files = sorted(f for f in os.listdir(d) if f.endswith(tag))
    if not files:
        return None, None, None
    try:
        j = json.load(open(os.path.join(d, files[-1])))
        st = j.get("collected_on") or {}
        now = _identity()
        stale = not st.get("source_sha256") or st["source_sha256"] != now["source_sha256"]
        return round(j["classes"][dom]["hbm_bytes_per_launch"]), "profiles/" + files[-1], stale
    except Exception:
        return None, None, None


_ID = None


def _identity():
    global _ID
    if _ID is None:
        from vlsat_amd import lib as _L
        _ID = _L.identity(_L.LIB_PATH)
    return _ID


def measure_traffic(argv, dom):
    """--measure-traffic: HBM bytes per launch of class `dom` measured NOW -- two rocprofv3 counter passes (FETCH_SIZE, then
    WRITE_SIZE: they do not fit one pass, and a counter pass never carries a trace domain besides --kernel-trace) over
    `bench.py --steps 3 --warmup 1 --no-cpu --no-profile --no-extra <the same workload arguments>`."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    tmp = tempfile.mkdtemp(prefix="vlsat_pmc_", dir="/tmp")
    got = {}
    for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        out = os.path.join(tmp, name)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "p", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu", "--no-profile", "--no-extra"] + argv
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
        if r.returncode:
            return None, f"rocprofv3 {ctr} pass failed: {r.stderr[-300:]}"
        import glob
        sub = os.path.dirname(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0])
        got[name] = pmc_summary.class_bytes(pmc_summary.load(sub), ctr)
    shutil.rmtree(tmp, ignore_errors=True)
    if dom not in got["fetch"] or dom not in got["write"]:
        return None, "class not in the counter output"
    return round(got["fetch"][dom] + got["write"][dom]), "measured in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"


def roofline_of(classes, mode, steps, falg, value, world, traffic=None, traffic_src=None, traffic_stale=None, traffic_measured=False):
    """`roofline` object of one timed run: the dominant kernel class by HIP-event time on the launch stream."""
    if not classes:
        return None
    dom = max(classes, key=lambda k: classes[k]["ms"])
    c = classes[dom]
    achieved = c["flops"] / (c["ms"] * 1e-3) / 1e12 if c["ms"] > 0 else 0.0
    # MFMAs per algorithmic product of the DOMINANT class.  Mode 4 (bf16x3_attn1) mixes both: its edge attention and the three
    # projections around it (q, k|v, out: 2 x 512 x 2048 of the 2 x 512 x (2 x 2560 + 2048 + heads) flops an edge row costs per layer)
    # run single-rounded, everything else split -- the attention class is priced against the full 2.5 PF, the GEMM class against the
    # flop-weighted blend (2.5 PF / 3 for whichever class dominated overstated the fraction by up to 3x)
    mfma_per_product = {"fp32": 1.0, "bf16": 1.0, "bf16_mixed": 1.0, "bf16x3": 3.0, "fp16_mixed": 1.0}.get(mode)
    note = {"fp32": "v_mfma_f32_32x32x2_f32", "bf16x3": "2.5 PF bf16 dense / 3 MFMAs per product", "bf16": "2.5 PF bf16 dense",
            "bf16_mixed": "2.5 PF bf16 dense", "fp16_mixed": "2.5 PF f16 dense (the bf16 rate on CDNA4)"}.get(mode)
    if mode == "bf16x3_attn1":
        if dom.startswith("flash"):
            mfma_per_product, note = 1.0, "2.5 PF bf16 dense (mode 4: the edge attention runs single-rounded)"
        else:
            single = 2.0 * 512 * 2048
            split = 2 * 2.0 * 512 * 2560 + (2 * 2.0 * (512 * 512 + 512 * 256 + 256 * 26) + 2 * 2.0 * (64 * 128 + 128 * 512)) / 3.0   # (heads and encoders: once per forward, ~ per 3 layers)
            f1 = single / (single + split)
            mfma_per_product = 3.0 - 2.0 * f1
            note = (f"2.5 PF bf16 dense / {mfma_per_product:.2f} MFMAs per product: the flop-weighted blend of this class in mode 4 "
                    f"({100 * f1:.0f} % of its flops -- the q / k|v / out projections of the edge attention -- are single-rounded, the rest split-bf16)")
    base = PEAK_FP32_MFMA_TFLOPS if mode == "fp32" else PEAK_BF16_MFMA_TFLOPS
    sust = SUSTAINED_FP32_MFMA_TFLOPS if mode == "fp32" else SUSTAINED_BF16_MFMA_TFLOPS
    peak = base / mfma_per_product
    total_ms = max(sum(x["ms"] for x in classes.values()), 1e-9)
    return {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": round(peak, 1),
            "peak_note": note,
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "peak_sustained": round(sust / mfma_per_product, 1),
            "frac_of_sustained": round(achieved / (sust / mfma_per_product), 4),
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src, "traffic_measured": bool(traffic_measured),
            "traffic_stale": None if traffic is None else (False if traffic_measured else bool(traffic_stale)),
            "launches_per_step": c["launches"] // steps, "avg_launch_ms": round(c["ms"] / max(c["launches"], 1), 4),
            "flop_per_launch": c["flops"] / max(c["launches"], 1),
            "whole_forward_tflops": round(falg * value / world / 1e12, 2),
            "whole_forward_frac": round(falg * value / world / 1e12 / peak, 4),
            "time_share": {k: round(v["ms"] / total_ms, 4) for k, v in classes.items()},
            "class_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)
                             for k, v in classes.items() if v["ms"] > 0 and v["flops"] > 0}}


def timed_run(model, d, n_scenes, steps, warmup, prof, dev):
    """W untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides; the wall time is the max
    over ranks.  A step = forward over this rank's batch + the one all-reduce of the metrics vector.  Per-step durations
    (HIP events on the launch stream, this rank) give the median SURVEY 8(d) asks for next to the mean."""
    def step():
        out = model(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
        return out, vdist.allreduce_metrics(vdist.scene_metrics(out, n_scenes))

    for _ in range(warmup):
        out, metrics = step()
    torch.cuda.synchronize()
    if prof:
        model.profile_enable(True)
        model.profile_read()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    vdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(steps):
        out, metrics = step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    local_dt = time.perf_counter() - t0             # this rank's own time (before it waits for the others)
    vdist.barrier()
    dt = vdist.max_over_ranks(time.perf_counter() - t0, dev)
    classes = model.profile_read() if prof else {}
    model.profile_enable(False)
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    median = per_step[steps // 2] if steps % 2 else 0.5 * (per_step[steps // 2 - 1] + per_step[steps // 2])
    return {"dt": dt, "local_ms": local_dt / steps * 1e3, "median_ms": median, "min_ms": per_step[0], "max_ms": per_step[-1],
            "classes": classes, "out": out, "metrics": metrics}


def two_in_flight(model, d, n_scenes, steps, warmup, dev):
    """The same steps with TWO of them in flight: step i on handle i % 2 (the model and one replica: own plans and scratch) on
    its own caller stream, so the end of a step -- where only the 2D edge lane still has work (profiles/r06_probes/lanes_bf16_mixed.txt)
    -- runs under the start of the next.  Every step is a complete forward + checksums of its own; reported NEXT to `value`, never as it."""
    models = [model] + model.replicas(1)
    streams = [torch.cuda.Stream(device=dev) for _ in models]

    def run(k):
        outs = [None, None]
        for i in range(k):
            with torch.cuda.stream(streams[i % 2]):
                out = models[i % 2](d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
                outs[i % 2] = (out, vdist.scene_metrics(out, n_scenes))
        return outs

    torch.cuda.synchronize()                      # the inputs were produced on the caller's stream
    run(2 * max(2, warmup))                      # (the replica's plan and scratch are built in its first step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    return {"what": "two steps in flight: step i on handle i % 2 (model + one replica), each on its own stream; every step a complete forward + checksums",
            "value": round(n_scenes * steps / dt, 2), "unit": "scenes/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "outputs_of_both_handles_bit_identical": bool(same)}


def eval_leg(model, scenes, d, n_obj, dev):
    """Untimed: the step AFTER the path on this rank's batch with seeded synthetic labels -- forward +
    GPU ranking + the additive counts vector (len(evaluate.fields()) = 361 counts), all-reduced once (evaluate.validation) -- so the line carries real evaluation
    metrics next to the checksums.  Labels are random: the accuracies are chance level by construction."""
    import numpy as np
    from vlsat_amd import evaluate as EV
    gts, rels = [], []
    for s in scenes:
        g = np.random.default_rng([1000 + s, 77])
        gts.append(g.integers(0, 160, n_obj))
        rels.append((g.random((n_obj * (n_obj - 1), 26)) < 0.04).astype(np.int64))
    b = {"obj_points": d["obj_points"], "obj_2d_feats": d["obj_2d_feats"], "descriptor": d["descriptor"],
         "batch_ids": d["batch_ids"], "edge_indices": d["edge_indices"].t().contiguous(),
         "gt_class": torch.from_numpy(np.concatenate(gts)).to(dev), "gt_rel_cls": torch.from_numpy(np.concatenate(rels)).to(dev)}
    b["fc_sizes"] = [n_obj] * len(scenes)               # (the loader's hint: the plan is keyed without reading the edge list back)
    reps = 5

    def leg(**kw):
        """first call (plans, buffers, kernels cold), then the mean of `reps` further calls; every call = forward, ranking, counts,
        one all-reduce, one host read of the summary"""
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = EV.validation(model, [b], dev, **kw)
        torch.cuda.synchronize()
        cold = (time.perf_counter() - t0) * 1e3
        vdist.barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            again = EV.validation(model, [b], dev, **kw)
        torch.cuda.synchronize()
        warm = vdist.max_over_ranks(time.perf_counter() - t1, dev) * 1e3 / reps
        assert all(float(again[k]) == float(first[k]) for k in first), "evaluation is not reproducible call to call"
        return first, cold, warm

    # the step as an evaluation LOOP runs it: ranks and counts stay on the device (vlsat_process_val_counts), one read at the end
    summ, cold_ms, warm_ms = leg(workers=1)
    # the reference-compatible form of the same step: numpy rank lists per batch on the host like Mmgnet.process_val, host counting
    summ_h, cold_h, warm_h = leg()
    assert all(float(summ_h[k]) == float(summ[k]) for k in summ), "device counts and host rank lists disagree"
    # ... and as a loop over a dataset runs it: 8 such batches per call of evaluate.validation -- one all-reduce and one host read for all of
    # them -- with one and with two batches in flight (workers = 2: a replica of the model on a second stream and host thread)
    loop = {}
    for w in (1, 2):
        EV.validation(model, [b] * 4, dev, workers=w)                 # (replica, plans and scratch of this worker count warm)
        torch.cuda.synchronize()
        vdist.barrier()
        t2 = time.perf_counter()
        many = EV.validation(model, [b] * 8, dev, workers=w)
        torch.cuda.synchronize()
        dt8 = vdist.max_over_ranks(time.perf_counter() - t2, dev)
        assert all(abs(float(many[k]) - float(summ[k])) < 1e-9 for k in summ if k != "scenes"), "the loop's percentages differ from one batch's"
        loop[f"workers_{w}"] = {"scenes_per_s_per_gpu": round(8 * len(scenes) / dt8, 1), "ms_per_batch": round(dt8 / 8 * 1e3, 2)}
    keep = ("scenes", "obj_acc@1_3d", "obj_acc@5_3d", "rel_acc@1_3d", "rel_acc@3_3d", "tri_acc@50_3d", "tri_acc@100_3d",
            "mean_recall@50_3d", "obj_acc@1_2d", "rel_acc@1_2d", "tri_acc@50_2d", "mean_recall@50_2d")
    return {"what": "forward + GPU ranking + the additive counts vector of evaluate.validation (device-side: vlsat_process_val_counts), one "
                    "all-reduce, one host read; synthetic random labels (chance-level accuracies), outside the timed region",
            "n_counts": len(EV.fields()), "ms_forward_plus_ranking_first_call": round(cold_ms, 2),
            "ms_forward_plus_ranking": round(warm_ms, 2), "scenes_per_s_per_gpu": round(len(scenes) / warm_ms * 1e3, 1),
            "steady_state": f"mean of {reps} further calls of evaluate.validation(workers=1) on the same batch (every call: forward, ranking, "
                            "counts, one all-reduce, one host read of the summary)",
            "reference_compatible_rank_lists": {"what": "the same step returning numpy rank lists per batch like Mmgnet.process_val "
                                                        "(evaluate.validation(workers=0)): identical summary, host-side counting",
                                                "ms_forward_plus_ranking": round(warm_h, 2),
                                                "scenes_per_s_per_gpu": round(len(scenes) / warm_h * 1e3, 1)},
            "loop_of_8_batches": dict(loop, what="evaluate.validation over 8 such batches in one call (device counts; one all-reduce and one host "
                                                  "read at the end), 1 / 2 batches in flight"),
            "metrics": {k: round(float(summ[k]), 4) for k in keep if k in summ}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scenes", type=int, default=64, help="scenes per GPU per step")
    ap.add_argument("--objects", type=int, default=40)
    ap.add_argument("--points", type=int, default=256)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--heads", type=int, default=8, help="MODEL.NUM_HEADS (4 | 8 | 16; not the headline configuration unless 8)")
    ap.add_argument("--dim-atten", type=int, default=256, help="MODEL.DIM_ATTEN (128 | 256 | 512)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="headline configuration only: skip the extra_configs legs (BASELINE configs[2] modes and configs[4]) "
                         "and the evaluation leg")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="measure roofline.traffic in this run: two extra rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same "
                         "workload on 3 steps (< 1 min).  This is the DEFAULT at --gpus 1 when rocprofv3 is on the box and the run is the "
                         "full one (no --no-extra); --no-measure-traffic quotes the committed profiles/ summary instead, labelled "
                         "traffic_measured=false and traffic_stale=<collected on another build>")
    ap.add_argument("--no-measure-traffic", action="store_true", help="never run the counter passes (see --measure-traffic)")
    ap.add_argument("--native-allreduce", action="store_true",
                    help="sum the metrics vector with the library's own RCCL entry point (vlsat_metrics_allreduce) "
                         "instead of torch.distributed.all_reduce")
    ap.add_argument("--debug-option", action="append", default=[], metavar="NAME=VALUE",
                    help="experiment switch of the library (vlsat_debug_option), e.g. node_attn_split=0; repeatable")
    ap.add_argument("--gemm-precision", default="fp32", choices=["fp32", "bf16x3", "bf16_mixed", "bf16", "bf16x3_attn1", "fp16_mixed"],
                    help="fp32 = BASELINE configs[1] (default, the headline); bf16x3 / bf16_mixed = configs[2] "
                         "(split-bf16 MFMA: <=1e-3; mixed single/split bf16: <=1e-2)")
    ap.add_argument("--lib", default="", help="another build of libvlsat_hip.so to load instead of the in-tree one (same-box A/B of a kernel change)")
    args = ap.parse_args()
    if args.lib:
        from vlsat_amd import lib as _L
        _L.LIB_PATH = os.path.abspath(args.lib)

    rank, local, world = vdist.init()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no MI355X visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    local = local % torch.cuda.device_count()          # (a 2-rank gloo dry run may share one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.native_allreduce:
        vdist.use_native_allreduce(rank, world)
    from vlsat_amd.model import VLSATModel

    cfg = VLSATConfig(N_LAYERS=args.layers, NUM_HEADS=args.heads, DIM_ATTEN=args.dim_atten)
    model = VLSATModel(cfg, str(dev)).load_state(synth.make_weights(cfg)).eval()
    model.set_gemm_precision(args.gemm_precision)
    for kv in args.debug_option:
        model.debug_option(kv.split("=")[0], int(kv.split("=")[1]))
    # this rank's shard of the global scene list (weak scaling: args.scenes per GPU)
    scenes = vdist.shard(args.scenes * world, rank, world)
    batch = synth.collate([synth.make_scene(args.objects, args.points, 1000 + s) for s in scenes])
    d = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    n_scenes = len(scenes)
    prof = not args.no_profile

    run = timed_run(model, d, n_scenes, args.steps, args.warmup, prof, dev)
    dt, out, metrics, classes = run["dt"], run["out"], run["metrics"], run["classes"]
    rank_ms = vdist.minmax_over_ranks(run["local_ms"], dev)
    default_wl = (args.scenes, args.objects, args.points, args.layers, args.heads, args.dim_atten) == (64, 40, 256, 3, 8, 256)
    evaluation, plain = None, None
    if not args.no_extra:                                # collectives: every rank takes part
        if prof:
            # The same K steps again without the per-class HIP events (both runs execute the 2D twin stages on the library's
            # second stream; the instrumented one, the headline `value`, records an event per kernel-class change on either
            # stream and charges a class the union of its intervals).
            r2 = timed_run(model, d, n_scenes, args.steps, args.warmup, False, dev)
            plain = {"what": "same batch and steps without the per-class HIP events (cost of the instrumentation: ~150 event records per step)",
                     "value": None, "ms_per_step": round(r2["dt"] / args.steps * 1e3, 3), "median_ms_per_step": round(r2["median_ms"], 3),
                     "_dt": r2["dt"]}
        evaluation = eval_leg(model, list(scenes), d, args.objects, dev)

    if rank != 0:
        return
    total_scenes = float(metrics[0].item()) if world > 1 else n_scenes
    value = total_scenes * args.steps / dt
    if plain:
        plain["value"] = round(total_scenes * args.steps / plain.pop("_dt"), 2)
    e_scene = args.objects * (args.objects - 1)
    falg = f_alg(args.objects, args.points, e_scene, args.layers)

    roofline = None
    if classes:
        dom = max(classes, key=lambda k: classes[k]["ms"])
        traffic, traffic_src, measured, stale = None, None, False, None
        import shutil as _sh
        want_measure = args.measure_traffic or (not args.no_measure_traffic and not args.no_extra and _sh.which("rocprofv3") is not None)
        if want_measure and world == 1:
            # the counter passes of this very workload, now (minutes: opt-in; the driver's plain run reads the committed summary)
            wl = ["--scenes", str(args.scenes), "--objects", str(args.objects), "--points", str(args.points), "--layers", str(args.layers),
                  "--heads", str(args.heads), "--dim-atten", str(args.dim_atten), "--gemm-precision", args.gemm_precision]
            traffic, traffic_src = measure_traffic(wl, dom)
            measured = traffic is not None
        if traffic is None and default_wl:
            traffic, traffic_src, stale = committed_traffic("cfg2", args.gemm_precision, dom)
        roofline = roofline_of(classes, args.gemm_precision, args.steps, falg, value, world, traffic, traffic_src, stale, measured)

    cpu, err, ref = None, None, None
    if world == 1 and not args.no_cpu:
        cpu, _ = cpu_baseline(cfg, args.objects, args.points)
        ref = oracle_all_scenes(cfg, batch, cpu["cores"])
        err = {k: float((g.cpu() - r).abs().max()) for k, g, r in zip(("obj3d", "obj2d", "rel3d", "rel2d"), out, ref)}
        err["scenes_checked"] = f"{n_scenes}/{n_scenes}"

    # ---- BASELINE configs[2] (bf16 matrix cores, same batch) and configs[4] (200 x 1024 stress scene) in the same line ----
    extra = []
    if world == 1 and not args.no_extra and default_wl and args.gemm_precision == "fp32":
        names = ("obj3d", "obj2d", "rel3d", "rel2d")

        def dom_traffic(workload, mode, cl):             # committed PMC summary of that configuration (not measured in this run)
            return committed_traffic(workload, mode, max(cl, key=lambda k: cl[k]["ms"])) if cl else (None, None, None)
        def two_runs(batch_d, scenes_n, steps, warmup):
            """An extra configuration is timed WITHOUT events on the schedule it ships with (in the bf16 modes the dependency-exact
            three-lane one: `value`), and profiled per kernel class on ONE stream (`prof_dual` = 0), so that a class's time is its
            kernels' own -- on three lanes the union of a class's intervals also holds what its launches wait for."""
            r = timed_run(model, batch_d, scenes_n, steps, warmup, False, dev)
            cl = {}
            if prof:
                model.debug_option("prof_dual", 0)
                cl = timed_run(model, batch_d, scenes_n, steps, warmup, True, dev)["classes"]
                model.debug_option("prof_dual", 1)
            r["classes"] = cl
            return r
        for mode in ("bf16x3", "bf16x3_attn1", "bf16_mixed", "fp16_mixed"):
            model.set_gemm_precision(mode)
            r = two_runs(d, n_scenes, args.steps, args.warmup)
            v = n_scenes * args.steps / r["dt"]
            e = None
            if ref is not None:
                e = {k: float((g.cpu() - x).abs().max()) for k, g, x in zip(names, r["out"], ref)}
                e["scenes_checked"] = f"{n_scenes}/{n_scenes}"
            ev_mode, pipelined = None, None
            if mode in ("bf16x3", "bf16_mixed"):
                pipelined = two_in_flight(model, d, n_scenes, args.steps, args.warmup, dev)
            if mode == "bf16_mixed":             # the step after the path behind the fastest forward (there the ranking weighs most)
                ev = eval_leg(model, list(scenes), d, args.objects, dev)
                ev_mode = {k: ev[k] for k in ("what", "ms_forward_plus_ranking", "scenes_per_s_per_gpu", "reference_compatible_rank_lists", "loop_of_8_batches")}
            extra.append({"workload": f"BASELINE configs[2]{' on fp16 instead of bf16' if mode == 'fp16_mixed' else ''}: 64 scenes x 40 objects x 256 pts, L=3, {mode}", "dtype": MODE_DTYPE[mode], "evaluation": ev_mode, "two_steps_in_flight": pipelined,
                          "timing": "value: steps without events on the shipped schedule; roofline: the same steps profiled on one stream",
                          "tolerance": 1e-2, "value": round(v, 2), "unit": "scenes/s", "ms_per_step": round(r["dt"] / args.steps * 1e3, 3),
                          "median_ms_per_step": round(r["median_ms"], 3), "steps": args.steps, "max_abs_err_vs_cpu_oracle": e,
                          "roofline": roofline_of(r["classes"], mode, args.steps, falg, v, 1, *dom_traffic("cfg2", mode, r["classes"]))})
        big = synth.make_batch(1, 200, 1024, seed0=5000)
        db = {k: torch.from_numpy(v).to(dev) for k, v in big.items()}
        falg5 = f_alg(200, 1024, 200 * 199, args.layers)
        gold = os.path.join(ROOT, "tests", "golden", "cfg5_n200_p1024_l3_sub.npz")
        for mode in ("fp32", "bf16_mixed", "fp16_mixed"):
            model.set_gemm_precision(mode)
            r = two_runs(db, 1, 5, 2)
            v = 5 / r["dt"]
            e = None
            if os.path.exists(gold):                 # committed oracle subsample of this very scene (tests/golden/make_golden_cfg5.py)
                import numpy as np
                z = np.load(gold)
                idx = torch.from_numpy(z["edge_idx"])
                got = [r["out"][0].cpu(), r["out"][1].cpu(), r["out"][2].cpu()[idx], r["out"][3].cpu()[idx]]
                e = {k: float((g - torch.from_numpy(z[k])).abs().max()) for k, g in zip(names, got)}
                e["checked"] = f"all 200 objects, {len(idx)} of 39800 edges (committed oracle subsample)"
            extra.append({"workload": f"BASELINE configs[4]: 1 scene x 200 objects x 1024 pts, dense graph E=39800, L=3, {mode}",
                          "dtype": MODE_DTYPE[mode], "tolerance": 1e-3 if mode == "fp32" else 1e-2, "value": round(v, 2), "unit": "scenes/s",
                          "ms_per_step": round(r["dt"] / 5 * 1e3, 3), "median_ms_per_step": round(r["median_ms"], 3), "steps": 5,
                          "max_abs_err_vs_cpu_oracle": e, "flop_per_scene_alg": falg5,
                          "roofline": roofline_of(r["classes"], mode, 5, falg5, v, 1, *dom_traffic("cfg5", mode, r["classes"]))})
        model.set_gemm_precision(args.gemm_precision)

    line = {
        "metric": "scenes/sec (3RScan-shaped, N=40 obj x 256 pts)", "value": round(value, 2), "unit": "scenes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "median_ms_per_step": round(run["median_ms"], 3), "step_ms_min_max": [round(run["min_ms"], 3), round(run["max_ms"], 3)],
        "rank_ms_per_step_min_max": [round(rank_ms[0], 3), round(rank_ms[1], 3)],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": MODE_DTYPE[args.gemm_precision],
        "data": "synthetic",
        "config": {"workload": f"{('BASELINE configs[1]' if args.gemm_precision == 'fp32' else 'BASELINE configs[2]') if default_wl else 'custom'}: batch of {args.scenes} synthetic scenes per GPU, "
                               f"{args.objects} objects x {args.points} pts, fully-connected edges "
                               f"(E={e_scene}/scene), {args.layers} GNN layers, {args.gemm_precision}"
                               + ("" if (args.heads, args.dim_atten) == (8, 256) else f", NUM_HEADS={args.heads}, DIM_ATTEN={args.dim_atten}"),
                   "scenes_per_gpu": args.scenes, "parallelism": f"scene-sharded x{world}"},
        "flop_per_scene_alg": falg,
        "metrics_allreduced": {k: float(v) for k, v in zip(vdist.METRIC_FIELDS, metrics.tolist())},
        "allreduce": "vlsat_metrics_allreduce (RCCL via the C ABI)" if args.native_allreduce else "torch.distributed.all_reduce",
        "uninstrumented": plain,
        "evaluation": evaluation,
        "library": _identity(),
        "roofline": roofline, "cpu_baseline": cpu, "max_abs_err_vs_cpu_oracle": err,
        "speedup_vs_cpu": round(value / cpu["value"], 1) if cpu else None,
        "extra_configs": extra or None,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
