#!/usr/bin/env python3
"""Per-kernel PMC summary from rocprofv3 --pmc CSV output (counter_collection + kernel_trace).
    python tools/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write out.md
Derived columns per kernel class (averages per launch over vlsat kernels only):
  MfmaUtil   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * 2.4 GHz): matrix-pipe busy time
               as a fraction of the NOMINAL clock, i.e. directly comparable with achieved/peak TFLOP/s.
               (The shader clock under this load is ~1.85 GHz by s_memtime vs s_memrealtime --
               DESIGN.md §5 -- so the pipe is busier than this number says.)
  HBM bytes  = 2 * FETCH_SIZE KB (gfx950 halves wide coalesced reads: MI355X_MICROARCH.md §HBM) and WRITE_SIZE KB as is
"""
import collections
import csv
import glob
import os
import sys


def stamp():
    """Identity of the library this summary was collected on (vlsat_amd.lib.identity) + the commit when a .git is around."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, root)
    try:
        import vlsat_amd  # noqa: F401
        from vlsat_amd import lib as L
        st = L.identity()
    except Exception as ex:      # (never let the stamp break a summary)
        st = {"error": repr(ex)}
    try:
        st["git_head"] = subprocess.run(["git", "-C", root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        st["git_head"] = None
    return st


def load(d):
    cc = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    kt = glob.glob(os.path.join(d, "*kernel_trace.csv"))[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if "vlsat::" not in name:
            continue
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name].add(r["Dispatch_Id"])
    for name, ids in cnt.items():
        per[name]["_launches"] = len(ids)
        per[name]["_dur_ns"] = sum(dur[i][0] for i in ids if i in dur)
    return per


CLASS_OF = {"layernorm512": "layernorm512", "row_invnorm512": "misc", "desc_tail": "misc", "edge_embed": "misc",
            "dist_bias": "misc", "gemm_ring": "gemm_f32", "gemm_p8": "gemm_f32", "gemm_splitk": "gemm_f32", "flash_attn_bf16": "flash_attn_f32",
            "flash_merge": "flash_attn_f32", "pointnet_bf16": "pointnet", "edge_gate_bf16": "edge_gate", "edge_gate_hd": "edge_gate",
            "edge_gate_bf16_hd": "edge_gate", "edge_gate_generic": "edge_gate", "node_attn_split": "node_attn"}     # bench.py's class names


def class_of(name):
    import re
    m = re.search(r"vlsat::(\w+?)(_kernel)?(<|$)", name)
    key = m.group(1) if m else name
    return CLASS_OF.get(key, key)


def class_bytes(per, counter):
    """bytes per launch of every kernel CLASS from one loaded counter pass (FETCH_SIZE is doubled: gfx950 tallies 128-B requests as 64 B)"""
    tot, n = collections.defaultdict(float), collections.defaultdict(float)
    k = 2.0 if counter == "FETCH_SIZE" else 1.0
    for name, c in per.items():
        tot[class_of(name)] += k * c[counter] * 1024
        n[class_of(name)] += c["_launches"]
    return {c: tot[c] / max(n[c], 1) for c in tot}


def main(sq, fetch, write, out, l2=None):
    a, f, w = load(sq), load(fetch), load(write)
    h = load(l2) if l2 and os.path.isdir(l2) else {}
    lines = ["| kernel | launches | avg us | MfmaUtil % (vs 2.4 GHz) | wave cycles: waiting / issue-stalled / issuing % | LDS bank-conflict % | HBM read MB/launch (2x FETCH_SIZE) | HBM write MB/launch | HBM GB/s | L2 hit % |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    for name in sorted(a, key=lambda k: -a[k]["_dur_ns"]):
        c = a[name]
        n = c["_launches"]
        us = c["_dur_ns"] / n / 1e3
        mf = 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024 * c["_dur_ns"] * 2.4, 1)
        bc = 100 * c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1)
        rd = 2 * f[name]["FETCH_SIZE"] * 1024 / max(f[name]["_launches"], 1) / 1e6 if name in f else float("nan")
        wr = w[name]["WRITE_SIZE"] * 1024 / max(w[name]["_launches"], 1) / 1e6 if name in w else float("nan")
        bw = (rd + wr) * 1e6 / (us * 1e-6) / 1e9 if us > 0 else 0
        wc = max(c.get("SQ_WAVE_CYCLES", 0), 1)
        stall = f"{100 * c.get('SQ_WAIT_ANY', 0) / wc:.0f} / {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} / {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.0f}"
        hit = 100 * h[name]["TCC_HIT"] / max(h[name]["TCC_HIT"] + h[name]["TCC_MISS"], 1) if name in h else float("nan")
        lines.append(f"| `{name}` | {int(n)} | {us:.1f} | {mf:.1f} | {stall} | {bc:.1f} | {rd:.1f} | {wr:.1f} | {bw:.0f} | {hit:.0f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # per kernel CLASS (all template instantiations together): HBM bytes per launch for bench.py's roofline.traffic
    import json
    cls = {}
    for name in a:
        key = class_of(name)
        d = cls.setdefault(key, {"launches": 0, "hbm_read_bytes": 0.0, "hbm_write_bytes": 0.0, "kernel_ns": 0.0})
        d["launches"] += int(a[name]["_launches"])
        d["kernel_ns"] += a[name]["_dur_ns"]
        if name in f:
            d["hbm_read_bytes"] += 2 * f[name]["FETCH_SIZE"] * 1024 * a[name]["_launches"] / max(f[name]["_launches"], 1)
        if name in w:
            d["hbm_write_bytes"] += w[name]["WRITE_SIZE"] * 1024 * a[name]["_launches"] / max(w[name]["_launches"], 1)
    for d in cls.values():
        d["hbm_bytes_per_launch"] = (d["hbm_read_bytes"] + d["hbm_write_bytes"]) / max(d["launches"], 1)
    json.dump({"collected_on": stamp(),
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 "
                       "--no-cpu --no-profile`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B "
                       "requests as 64 B); WRITE_SIZE as reported (uncalibrated)", "classes": cls},
              open(out.replace(".md", ".json"), "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:6])
