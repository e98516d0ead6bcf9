#!/bin/bash
# round 6: 256-query attention tiles from smaller scenes on ("flash_bq_big_min"): step A/B at cfg 3
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_eighth
mkdir -p "$OUT"; cd "$ROOT"
one() { local mode=$1; shift; python bench.py --gemm-precision $mode --steps 40 --warmup 5 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2 3; do
  for mode in bf16_mixed bf16x3_attn1; do
    for v in 4096 1024; do echo "$mode flash_bq_big_min=$v rep=$rep $(one $mode --debug-option flash_bq_big_min=$v)" >> "$OUT/ab.txt"; done
  done
done
cat "$OUT/ab.txt"
timeout 600 python -m pytest tests/test_hip_round5.py tests/test_hip_forward.py -x -q -k "big or 256 or cfg3 or bq" 2>&1 | tail -3
