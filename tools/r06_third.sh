#!/bin/bash
# round 6, third GPU session: deduplicated gather init + K-tile rotation of the 8-phase GEMM: tests, kernel A/B, step A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_third
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "p8" > "$OUT/tests.txt" 2>&1
tail -5 "$OUT/tests.txt"
python tools/p8_check.py --ab --no-check --structured --cold 6 --iters 120 > "$OUT/p8_ab_cold.txt" 2>&1; cat "$OUT/p8_ab_cold.txt"
python tools/p8_check.py --ab --no-check --structured --iters 60 > "$OUT/p8_ab_warm.txt" 2>&1; cat "$OUT/p8_ab_warm.txt"
: > "$OUT/step_ab.txt"
for rep in 1 2 3; do
  for opt in "gemm_dedup=1" "gemm_dedup=0" "gemm_k_rot=1" "gemm_k_rot=2"; do
    v=$(python bench.py --gemm-precision bf16_mixed --steps 40 --warmup 5 --no-cpu --no-extra --no-profile --debug-option $opt 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])')
    echo "bf16_mixed $opt rep=$rep $v" >> "$OUT/step_ab.txt"
  done
done
cat "$OUT/step_ab.txt"
