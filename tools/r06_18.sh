#!/bin/bash
# 64 queries per wave in the half-row edge attention ("flash_qg" 0 | 1 | 2): bit-identity test, then interleaved A/B at the bench batch
# (128-query tiles, two waves per block) and at cfg 5 (256-query tiles, four waves per block)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_18
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_hip_round6.py -q -m gpu -k "64_queries" 2>&1 | tail -5 > "$OUT/test_qg.log"
cat "$OUT/test_qg.log"
ab() {  # label, extra args...
  local label=$1; shift
  for rep in 1 2; do for v in 0 1 2; do
    timeout 300 python bench.py --no-cpu --no-extra --steps 10 "$@" --debug-option flash_qg=$v 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$label flash_qg=$v: %.2f scenes/s, %.3f ms/step; flash: share %.4f, %.1f TF' % (d['value'], d['ms_per_step'], r['time_share'].get('flash', 0), r['class_tflops'].get('flash', 0)))"
  done; done
}
{
ab cfg3_bf16_mixed --gemm-precision bf16_mixed
ab cfg5_bf16_mixed --gemm-precision bf16_mixed --scenes 1 --objects 200 --points 1024
ab cfg3_bf16x3_attn1 --gemm-precision bf16x3_attn1
} 2>&1 | tee "$OUT/ab_flash_qg.txt"
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > "$OUT/tests_gpu.log"
tail -3 "$OUT/tests_gpu.log"
