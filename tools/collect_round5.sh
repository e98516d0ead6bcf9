#!/bin/bash
# Everything DESIGN.md quotes for round 5, in one GPU session; outputs under gpurun_out/r05/ (tools/publish_profiles.sh r05
# copies the summaries to profiles/).  PART=a|b|c runs a third of it (gpurun calls are time-limited).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05
PART=${PART:-abc}
mkdir -p "$OUT"
cd "$ROOT"
if [[ $PART == *a* ]]; then
python bench.py > "$OUT/bench_fp32.json" 2> "$OUT/bench_fp32.err"                       # the driver's command: headline + extra_configs
python bench.py --gemm-precision bf16x3 --no-extra > "$OUT/bench_cfg3_bf16x3.json" 2> "$OUT/bench_cfg3_bf16x3.err"
python bench.py --gemm-precision bf16_mixed --no-extra > "$OUT/bench_cfg3_bf16_mixed.json" 2> "$OUT/bench_cfg3_bf16_mixed.err"
tools/profile_run.sh r05/prof_fp32 --no-extra > /dev/null 2>&1
tools/profile_run.sh r05/prof_cfg3 --gemm-precision bf16x3 --no-extra > /dev/null 2>&1
tools/profile_run.sh r05/prof_cfg3_mixed --gemm-precision bf16_mixed --no-extra > /dev/null 2>&1
tools/profile_run.sh r05/prof_cfg3_attn1 --gemm-precision bf16x3_attn1 --no-extra > /dev/null 2>&1
fi
if [[ $PART == *b* ]]; then
tools/profile_run.sh r05/prof_cfg5_fp32 --scenes 1 --objects 200 --points 1024 --no-extra > /dev/null 2>&1
tools/profile_run.sh r05/prof_cfg5_mixed --scenes 1 --objects 200 --points 1024 --gemm-precision bf16_mixed --no-extra > /dev/null 2>&1
tools/launch_list.sh r05/launches_fp32 --debug-option dual_stream=0 > /dev/null 2>&1
tools/launch_list.sh r05/launches_bf16_mixed --gemm-precision bf16_mixed --debug-option dual_stream=0 > /dev/null 2>&1
python tools/latency_probe.py > "$OUT/latency_fp32.txt" 2>&1
python tools/latency_probe.py --gemm-precision bf16_mixed > "$OUT/latency_bf16_mixed.txt" 2>&1
python tools/val_loop_probe.py > "$OUT/val_loop_fp32.txt" 2>&1
python tools/val_loop_probe.py --gemm-precision bf16_mixed > "$OUT/val_loop_bf16_mixed.txt" 2>&1
python tools/val_loop_probe.py --objects 40 --workers 1,2,4,8 > "$OUT/val_loop_fp32_n40.txt" 2>&1
tools/ab_sched.sh 2 > "$OUT/ab_sched.txt" 2>&1
fi
if [[ $PART == *c* ]]; then
python tools/eval_synth.py > "$OUT/eval_synth.txt" 2>&1
python tools/stress_scan.py > "$OUT/stress_scan.txt" 2>&1
python tools/fuzz_forward.py --iters 120 > "$OUT/fuzz_forward.txt" 2>&1
python tools/soak_forward.py > "$OUT/soak_forward.txt" 2>&1
python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -12 > "$OUT/tests_gpu.log"
fi
du -sh "$OUT"; ls "$OUT"
