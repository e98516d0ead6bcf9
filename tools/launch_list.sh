#!/bin/bash
# Every kernel launch of ONE bench step, in launch order, with its duration (rocprofv3 kernel trace):
#   tools/launch_list.sh <tag> [bench.py arguments ...]   ->  gpurun_out/<tag>/launches.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o t -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu --no-extra --no-profile "$@" > "$OUT/bench.json" 2> "$OUT/kt.log"
python - "$OUT" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last pointnet launch on
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "pointnet" in n]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = n.replace("vlsat::", "")
    return n.split("(")[0][:70]
with open(out + "/launches.txt", "w") as o:
    prev_end = t0; busy = 0
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        o.write(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):5d}  {short(r['Kernel_Name'])}\n")
        prev_end = max(prev_end, e)
    o.write(f"# {len(rows) - i0} launches, busy {busy / 1e6:.3f} ms, span {(prev_end - t0) / 1e6:.3f} ms\n")
print(open(out + "/launches.txt").read()[-300:])
PY
rm -rf "$OUT/kt"
