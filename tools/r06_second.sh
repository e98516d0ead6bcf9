#!/bin/bash
# round 6, second GPU session: new ranking kernels + scan pinning on the device, eval step timing + kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_second
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_hip_metrics.py tests/test_hip_scan.py tests/test_hip_prep.py tests/test_hip_round5.py -x -q > "$OUT/tests.txt" 2>&1
tail -15 "$OUT/tests.txt"
python tools/metrics_bench.py > "$OUT/metrics_bench.txt" 2>&1; cat "$OUT/metrics_bench.txt"
python bench.py --steps 10 --warmup 3 --no-cpu --no-measure-traffic > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<'PY' "$OUT/bench.json"
import json,sys
j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print("value", j["value"], "eval", json.dumps(j["evaluation"], indent=1)[:1500])
for x in j["extra_configs"] or []: print(x["workload"][-30:], x["value"], x["ms_per_step"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt_eval" -o ev -- python "$ROOT/tools/metrics_bench.py" > /dev/null 2> "$OUT/kt_eval.log"
python "$ROOT/tools/rocprof_summary.py" "$OUT"/kt_eval/ev_results.db "$OUT/eval_kernel_stats.md" > /dev/null 2>> "$OUT/kt_eval.log"
grep -n "rank_kernel\|sort_probs\|softmax_rows\|eval_counts" "$OUT/eval_kernel_stats.md"
rm -rf "$OUT/kt_eval"
