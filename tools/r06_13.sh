#!/bin/bash
# round 6: remainder tiles of the big bf16 GEMMs as balanced rounds of the 8-phase launch instead of a tail launch ("gemm_p8_part_min")
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_13
mkdir -p "$OUT"; cd "$ROOT"
one() { local mode=$1 steps=$2; shift 2; python bench.py --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2 3; do
  for v in 0 24 12 6; do
    echo "bf16_mixed gemm_p8_part_min=$v rep=$rep $(one bf16_mixed 40 --debug-option gemm_p8_part_min=$v)" >> "$OUT/ab.txt"
  done
  for v in 0 24 12; do
    echo "bf16x3 gemm_p8_part_min=$v rep=$rep $(one bf16x3 25 --debug-option gemm_p8_part_min=$v)" >> "$OUT/ab.txt"
  done
done
cat "$OUT/ab.txt"
