#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db on ROCm 7.2) into the per-kernel stats table kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db profiles/r01_bench_kernel_stats.md
"""
import sqlite3
import sys


def main(db, out):
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
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
