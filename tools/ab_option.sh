#!/bin/bash
# A/B of a vlsat_debug_option on the bench batch:  tools/ab_option.sh gate_row_map "0 1" "fp32 bf16_mixed bf16x3" [class]
cd "$(dirname "$0")/.."
opt=$1; vals=${2:-"0 1"}; modes=${3:-"fp32 bf16_mixed bf16x3"}; cls=${4:-edge_gate}
for m in $modes; do for v in $vals; do
  timeout 300 python bench.py --no-cpu --no-extra --steps 10 --gemm-precision $m --debug-option $opt=$v 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$m $opt=$v: %.1f scenes/s, %.3f ms/step; $cls: share %.4f, %.1f TF' % (d['value'], d['ms_per_step'], r['time_share'].get('$cls', 0), r['class_tflops'].get('$cls', 0)))"
done; done
