#!/bin/bash
# Interleaved A/B of several builds of the library:  tools/ab_libs.sh "<lib1.so> <lib2.so> ..." [reps] [mode] [extra bench args]
# ("-" = the in-tree build)
cd "$(dirname "$0")/.."
libs=$1; reps=${2:-2}; m=${3:-bf16_mixed}; shift 3
for r in $(seq $reps); do for l in $libs; do
  libarg=""; [ "$l" != "-" ] && libarg="--lib $l"
  timeout 300 python bench.py --no-cpu --no-extra --steps 20 --gemm-precision $m $libarg "$@" 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline'] or {}
print('$m $l: %.1f scenes/s, %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['median_ms_per_step']), ' '.join('%s %.0f' % kv for kv in (r.get('class_tflops') or {}).items()))"
done; done
