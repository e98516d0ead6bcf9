#!/bin/bash
# One scene per call under rocprofv3 --kernel-trace: how much of the wall time per call is kernel execution, how much
# is the gap between dependent launches?   tools/single_scene_trace.sh <tag> [latency_probe.py arguments]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o probe -- python "$ROOT/tools/latency_probe.py" "$@" > "$OUT/latency.txt" 2> "$OUT/kt.log"
python - "$OUT" <<'PY'
import sqlite3, sys
out = sys.argv[1]
c = sqlite3.connect(out + "/kt/probe_results.db")
rows = list(c.execute("select start, end, name from kernels order by start"))
vl = [(s, e, n) for s, e, n in rows if "vlsat::" in n]
busy = sum(e - s for s, e, _ in vl)
# gaps between consecutive vlsat kernels that are shorter than 200 us (= inside one forward)
gaps = [vl[i + 1][0] - vl[i][1] for i in range(len(vl) - 1)]
inside = [g for g in gaps if 0 <= g < 200_000]
import collections
per = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in vl:
    k = n_.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    per[k][0] += 1
    per[k][1] += e_ - s_
top = sorted(per.items(), key=lambda kv: -kv[1][1])[:14]
tab = "\n".join(f"  {k:70s} {v[0]:6d} launches  {v[1] / v[0] / 1e3:7.1f} us avg  {100 * v[1] / busy:5.1f} %" for k, v in top)
txt = (f"{len(vl)} vlsat kernel launches: total execution {busy / 1e6:.2f} ms, mean {busy / len(vl) / 1e3:.2f} us per kernel; "
       f"gaps inside a forward: mean {sum(inside) / max(len(inside), 1) / 1e3:.2f} us, total {sum(inside) / 1e6:.2f} ms "
       f"({100 * sum(inside) / (busy + sum(inside)):.0f} % of busy + gaps)\n" + tab + "\n")
open(out + "/trace_summary.txt", "w").write(txt)
print(txt)
PY
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf "$OUT/kt"
cat "$OUT/latency.txt"
