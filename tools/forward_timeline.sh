#!/bin/bash
# Kernel-by-kernel timeline of ONE one-scene forward under rocprofv3 --kernel-trace:
#   tools/forward_timeline.sh <tag> <objects> [fp32|bf16x3|bf16_mixed|bf16] [points] [debug option NAME=VALUE] [output suffix]
# writes gpurun_out/<tag>/timeline_<objects>.txt: every kernel of the last of 6 identical calls with its start offset,
# duration and the gap to the previous kernel's end on the same stream view (two-stream plans overlap: start offsets tell).
set -u
TAG=$1; N=$2; PREC=${3:-fp32}; P=${4:-256}; OPT=${5:-}; SUF=${6:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cat > /tmp/_one_scene.py <<PY
import sys, torch, numpy as np
sys.path.insert(0, "$ROOT")
import vlsat_amd
from vlsat_amd import VLSATConfig, synth
from vlsat_amd.model import VLSATModel
cfg = VLSATConfig(N_LAYERS=3)
m = VLSATModel(cfg, "cuda:0").load_state(synth.make_weights(cfg)).eval().set_gemm_precision("$PREC")
opt = "$OPT"
if opt:
    m.debug_option(opt.split("=")[0], int(opt.split("=")[1]))
b = synth.collate([synth.make_scene($N, $P, 1)])
d = {k: torch.from_numpy(v).to("cuda:0") for k, v in b.items()}
for i in range(6):
    out = m(d["obj_points"], d["obj_2d_feats"], d["edge_indices"], d["descriptor"], d["batch_ids"])
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt_$N" -o one -- python /tmp/_one_scene.py > "$OUT/kt_$N.log" 2>&1
python - "$OUT" "$N" "$PREC$OPT" "$SUF" <<'PY'
import sqlite3, sys, glob
out, n, prec, suf = sys.argv[1:5]
db = glob.glob(f"{out}/kt_{n}/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = [(s, e, nm) for s, e, nm in c.execute("select start, end, name from kernels order by start") if "vlsat::" in nm]
# split into forwards: the script synchronises between calls -> the device idles for tens of microseconds
fw, cur = [], []
for r in rows:
    if cur and r[0] - max(x[1] for x in cur) > 40_000:
        fw.append(cur); cur = []
    cur.append(r)
fw.append(cur)
last = fw[-1]
t0 = last[0][0]
lines = [f"{n} objects, {prec}: {len(last)} kernels, first start -> last end {(max(x[1] for x in last) - t0) / 1e3:.1f} us, "
         f"sum of durations {sum(e - s for s, e, _ in last) / 1e3:.1f} us"]
prev_end = t0
for s, e, nm in last:
    k = nm.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace("vlsat::", "")
    lines.append(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {k}")
    prev_end = max(prev_end, e)
open(f"{out}/timeline_{n}{suf}.txt", "w").write("\n".join(lines) + "\n")
print(lines[0])
PY
rm -rf "$OUT/kt_$N"
