#!/usr/bin/env python3
"""Per-queue timeline of ONE bench step from a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -- python bench.py ...):
which hardware queue (lane) runs which kernel when, how long every lane is busy, and where the step's wall time has NO kernel
running on any lane or only a latency-bound one.   python tools/lane_timeline.py DIR/NAME_results.db [out.txt]"""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = next((k for k in ("queue_id", "queue", "stream_id", "stream") if k in cols), None)
    rows = list(c.execute(f"select start, end, name, {qcol or '0'} from kernels order by start"))
    rows = [r for r in rows if "vlsat::" in r[2]]
    starts = [i for i, r in enumerate(rows) if "pointnet" in r[2]]
    i0 = starts[-1]
    i1 = len(rows)
    step = rows[i0:i1]
    t0 = step[0][0]
    t1 = max(r[1] for r in step)
    lanes = sorted({r[3] for r in step})
    lines = [f"one step: {len(step)} kernels on {len(lanes)} queues, wall {(t1 - t0) / 1e3:.1f} us, sum of kernel durations {sum(r[1] - r[0] for r in step) / 1e3:.1f} us"]
    for q in lanes:
        k = [r for r in step if r[3] == q]
        lines.append(f"  queue {q}: {len(k)} kernels, busy {sum(r[1] - r[0] for r in k) / 1e3:.1f} us, first start {(k[0][0] - t0) / 1e3:.1f}, last end {(max(r[1] for r in k) - t0) / 1e3:.1f}")
    # coverage: time with 0 / 1 / 2 / 3 kernels running
    ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
    cov, cur, last = {}, 0, t0
    for t, d in ev:
        cov[cur] = cov.get(cur, 0) + (t - last)
        cur += d
        last = t
    lines.append("  wall time by number of kernels running: " + ", ".join(f"{n}: {v / 1e3:.1f} us" for n, v in sorted(cov.items())))

    def short(n):
        return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("vlsat::", "").split("(")[0][:60]
    for s, e, n, q in step:
        lines.append(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  q{q}  {short(n)}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print("\n".join(lines[:8]))


if __name__ == "__main__":
    main(*sys.argv[1:3])
