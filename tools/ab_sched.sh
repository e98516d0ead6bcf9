#!/bin/bash
# A/B of the forward schedule on the bench batch, variants interleaved in one session (boxes differ by +-5 %):
#   tools/ab_sched.sh [reps] [modes]     -> "sched=1" dependency-exact three lanes (round 5), "sched=0" fork / join (round 4),
#                                           "dual_stream=0" one stream
cd "$(dirname "$0")/.."
reps=${1:-2}; modes=${2:-"bf16_mixed bf16x3 fp32"}
for m in $modes; do for r in $(seq $reps); do for v in "sched=1" "sched=0" "dual_stream=0"; do for p in "" "--no-profile"; do
  timeout 300 python bench.py --no-cpu --no-extra --steps 20 --gemm-precision $m --debug-option $v $p 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline'] or {}
print('$m $v ${p:-profiled}: %.1f scenes/s, %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['median_ms_per_step']), ' '.join('%s %.0f' % kv for kv in (r.get('class_tflops') or {}).items()))"
done; done; done; done
