#!/bin/bash
# round 6: "gemm_p8_part_min" in the other precisions
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_14
mkdir -p "$OUT"; cd "$ROOT"
one() { local mode=$1 steps=$2; shift 2; python bench.py --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2 3; do
  for v in 0 48 24 12; do
    echo "fp32 gemm_p8_part_min=$v rep=$rep $(one fp32 15 --debug-option gemm_p8_part_min=$v)" >> "$OUT/ab.txt"
    echo "bf16x3_attn1 gemm_p8_part_min=$v rep=$rep $(one bf16x3_attn1 25 --debug-option gemm_p8_part_min=$v)" >> "$OUT/ab.txt"
  done
  echo "cfg5 bf16_mixed gemm_p8_part_min=0 rep=$rep $(one bf16_mixed 6 --scenes 1 --objects 200 --points 1024 --debug-option gemm_p8_part_min=0)" >> "$OUT/ab.txt"
  echo "cfg5 bf16_mixed gemm_p8_part_min=12 rep=$rep $(one bf16_mixed 6 --scenes 1 --objects 200 --points 1024 --debug-option gemm_p8_part_min=12)" >> "$OUT/ab.txt"
done
cat "$OUT/ab.txt"
