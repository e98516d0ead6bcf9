#!/bin/bash
# round 6, first GPU session: (a) scenes-per-step sweep, (b) kernel trace of the ranking step (item 3)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_first
mkdir -p "$OUT"
cd "$ROOT"
: > "$OUT/sweep.txt"
for rep in 1 2; do
for mode in bf16_mixed fp32 bf16x3; do
  for s in 8 16 32 64 128; do
    steps=$(( 1280 / s )); [ $steps -gt 60 ] && steps=60
    line=$(python bench.py --gemm-precision $mode --scenes $s --steps $steps --warmup 3 --no-cpu --no-extra --no-profile 2>>"$OUT/sweep.err" | tail -1)
    echo "$mode scenes=$s rep=$rep $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print("scenes/s", j["value"], "ms/step", j["ms_per_step"], "ms/scene", round(j["ms_per_step"]/j["config"]["scenes_per_gpu"],5))')" >> "$OUT/sweep.txt"
  done
done
done
cat "$OUT/sweep.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt_eval" -o ev -- python "$ROOT/tools/metrics_bench.py" > "$OUT/metrics_bench.txt" 2> "$OUT/kt_eval.log"
python "$ROOT/tools/rocprof_summary.py" "$OUT"/kt_eval/ev_results.db "$OUT/eval_kernel_stats.md" > /dev/null 2>> "$OUT/kt_eval.log"
cat "$OUT/metrics_bench.txt"
head -40 "$OUT/eval_kernel_stats.md"
rm -rf "$OUT/kt_eval"
