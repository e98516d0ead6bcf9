#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db on ROCm 7.2) into the per-kernel stats table kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db profiles/r01_bench_kernel_stats.md [bench.json]

With the bench line of the traced run as third argument the table ends with a footer that recomputes roofline.achieved / frac
from the trace (sum of GEMM kernel time / forwards against the GEMM flops of a forward).
"""
import os
import sqlite3
import sys


def footer(c, bench_json):
    """Reproduces roofline.achieved / frac of the bench line from this trace alone: sum of the GEMM kernels' durations over the
    forwards in the trace against the GEMM flops of one forward (2 M N K of every launch, summed by the library and printed in
    the bench line: roofline.flop_per_launch x launches_per_step).  Only meaningful for a SERIALISED trace (dual_stream=0):
    with two streams kernels of both overlap and their durations include waiting for CUs."""
    import json
    line = None
    for ln in open(bench_json):
        if ln.startswith("{"):
            line = json.loads(ln)
    if not line or not line.get("roofline"):
        return []
    r = line["roofline"]
    fwd = list(c.execute("select count(*) from kernels where name like '%pointnet%'"))[0][0]
    # kernels of the bench line's dominant class (bench.py roofline.kernel): GEMM launches, or the edge attention (+ its merge) at cfg 5
    pat = {"gemm_f32": "%gemm_%", "flash_attn_f32": "%flash_%", "pointnet": "%pointnet%", "edge_gate": "%edge_gate%"}.get(r["kernel"], "%gemm_%")
    gemm_ns = list(c.execute("select sum(duration) from kernels where name like ?", (pat,)))[0][0] or 0
    all_ns = list(c.execute("select sum(duration) from kernels where name like '%vlsat::%'"))[0][0] or 0
    if not fwd or not gemm_ns:
        return []
    flops = r["flop_per_launch"] * r["launches_per_step"]
    ms = gemm_ns / 1e6 / fwd
    tf = flops / (ms * 1e-3) / 1e12
    return ["", f"Footer -- the roofline of the bench line from this file alone ({fwd} forwards in the trace, kernel class `{r['kernel']}`):", "",
            "| sum of the class's kernel time | per forward | the class's flops per forward (summed over its launches by the library) | TFLOP/s | peak | frac | bench line (HIP events, same run) |",
            "|---|---|---|---|---|---|---|",
            f"| {gemm_ns / 1e6:.3f} ms | {ms:.3f} ms | {flops / 1e9:.2f} GFLOP | {tf:.1f} | {r['peak']} | {tf / r['peak']:.4f} | achieved {r['achieved']}, frac {r['frac']} |",
            "", f"All library kernels: {all_ns / 1e6 / fwd:.3f} ms per forward; the bench line's ms_per_step: {line['ms_per_step']}."]


def main(db, out, bench_json=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | "
                     f"{100 * tot / total:.2f} |")
    # per launch-geometry breakdown of the GEMM kernels (one row per distinct grid size)
    lines += ["", "GEMM launches by grid (blocks) -- one shape class per row:", "",
              "| kernel | blocks | calls | avg us |", "|---|---|---|---|"]
    for name, g, n, avg in c.execute("select name, grid_x / workgroup_x, count(*), avg(duration) from kernels "
                                     "where name like '%gemm_f32%' group by name, grid_x order by avg(duration) desc"):
        lines.append(f"| `{name.split('(')[0]}` | {g} | {n} | {avg / 1e3:.1f} |")
    if bench_json:
        lines += footer(c, bench_json)
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import pmc_summary
        st = pmc_summary.stamp()
        lines += ["", f"collected on: source_sha256 {st.get('source_sha256')}, lib_sha256 {st.get('lib_sha256')} ({st.get('lib_bytes')} B), git {st.get('git_head')}"]
    except Exception:
        pass
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14] + lines[-4:]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
