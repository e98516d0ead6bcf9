#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_seventh
mkdir -p "$OUT"; cd "$ROOT"
timeout 900 python -m pytest tests/test_hip_metrics.py tests/test_hip_round5.py tests/test_evaluate_cpu.py -x -q > "$OUT/tests.txt" 2>&1; tail -5 "$OUT/tests.txt"
python tools/metrics_bench.py 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o ev -- python "$ROOT/tools/metrics_bench.py" > /dev/null 2>&1
python "$ROOT/tools/rocprof_summary.py" "$OUT"/kt/ev_results.db "$OUT/eval_kernel_stats.md" > /dev/null 2>&1
grep -n "rank_kernel\|sort_probs\|softmax_rows" "$OUT/eval_kernel_stats.md" | cut -c1-160
rm -rf "$OUT/kt"
