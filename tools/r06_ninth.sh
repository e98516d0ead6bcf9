#!/bin/bash
# round 6: A/B of switches that exist already, on the final library: split-K for the tails / node rows, fused aggregation in fp32, schedules
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_ninth
mkdir -p "$OUT"; cd "$ROOT"
one() { local mode=$1 steps=$2; shift 2; python bench.py --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2 3; do
  for opt in "gemm_splitk=1" "gemm_splitk=0"; do echo "bf16_mixed $opt rep=$rep $(one bf16_mixed 40 --debug-option $opt)" >> "$OUT/ab.txt"; done
  for opt in "gemm_splitk=1" "gemm_splitk=0" "gate_fuse_agg=2" "sched=1"; do echo "fp32 $opt rep=$rep $(one fp32 15 --debug-option $opt)" >> "$OUT/ab.txt"; done
done
cat "$OUT/ab.txt"
