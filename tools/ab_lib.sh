#!/bin/bash
# Interleaved A/B of two builds of the library on the bench batch (uninstrumented steps + per-class profile of one run each):
#   tools/ab_lib.sh <other.so> [reps] [modes] [extra bench args]       ("new" = the in-tree build)
cd "$(dirname "$0")/.."
other=$1; reps=${2:-2}; modes=${3:-"bf16_mixed bf16x3 fp32"}; shift 3
for m in $modes; do for r in $(seq $reps); do for v in new old; do
  libarg=""; [ $v = old ] && libarg="--lib $other"
  timeout 300 python bench.py --no-cpu --no-extra --steps 20 --gemm-precision $m $libarg "$@" 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline'] or {}
print('$m $v: %.1f scenes/s, %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['median_ms_per_step']), ' '.join('%s %.0f' % kv for kv in (r.get('class_tflops') or {}).items()))"
done; done; done
