#!/bin/bash
# round 6: new remainder thresholds of the 8-phase GEMM as DEFAULTS: GPU suite, then every mode against the old rule
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_15
mkdir -p "$OUT"; cd "$ROOT"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3
one() { local mode=$1 steps=$2; shift 2; python bench.py --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2 3; do
  echo "bf16_mixed default rep=$rep $(one bf16_mixed 40)" >> "$OUT/ab.txt"
  echo "bf16_mixed old(32) rep=$rep $(one bf16_mixed 40 --debug-option gemm_p8_part_min=32)" >> "$OUT/ab.txt"
  echo "bf16x3 default rep=$rep $(one bf16x3 25)" >> "$OUT/ab.txt"
  echo "bf16x3 old(160) rep=$rep $(one bf16x3 25 --debug-option gemm_p8_part_min=160)" >> "$OUT/ab.txt"
  echo "bf16x3_attn1 default rep=$rep $(one bf16x3_attn1 25)" >> "$OUT/ab.txt"
  echo "bf16x3_attn1 old(160) rep=$rep $(one bf16x3_attn1 25 --debug-option gemm_p8_part_min=160)" >> "$OUT/ab.txt"
  echo "fp32 default rep=$rep $(one fp32 15)" >> "$OUT/ab.txt"
done
cat "$OUT/ab.txt"
