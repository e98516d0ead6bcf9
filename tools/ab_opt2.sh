#!/bin/bash
# Interleaved A/B of one vlsat_debug_option on the bench batch, uninstrumented steps:
#   tools/ab_opt2.sh <option> "<values>" [reps] [modes] [extra bench args]
cd "$(dirname "$0")/.."
opt=$1; vals=${2:-"1 0"}; reps=${3:-2}; modes=${4:-"bf16_mixed bf16x3 fp32"}; shift 4
for m in $modes; do for r in $(seq $reps); do for v in $vals; do
  timeout 300 python bench.py --no-cpu --no-extra --no-profile --steps 20 --gemm-precision $m --debug-option $opt=$v "$@" 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$m $opt=$v: %.1f scenes/s, %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['median_ms_per_step']))"
done; done; done
