#!/bin/bash
# Everything DESIGN.md quotes for round 3, in one GPU session; outputs under gpurun_out/r03/ (tools/publish_profiles.sh r03
# copies the summaries to profiles/).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r03
mkdir -p "$OUT"
cd "$ROOT"
python bench.py > "$OUT/bench_fp32.json" 2> "$OUT/bench_fp32.err"                       # the driver's command: headline + extra_configs
python bench.py --gemm-precision bf16x3 --no-extra > "$OUT/bench_cfg3_bf16x3.json" 2> "$OUT/bench_cfg3_bf16x3.err"
python bench.py --gemm-precision bf16_mixed --no-extra > "$OUT/bench_cfg3_bf16_mixed.json" 2> "$OUT/bench_cfg3_bf16_mixed.err"
tools/profile_run.sh r03/prof_fp32 --no-extra > /dev/null 2>&1
tools/profile_run.sh r03/prof_cfg3 --gemm-precision bf16x3 --no-extra > /dev/null 2>&1
tools/profile_run.sh r03/prof_cfg3_mixed --gemm-precision bf16_mixed --no-extra > /dev/null 2>&1
tools/profile_run.sh r03/prof_heads4 --heads 4 --no-extra > /dev/null 2>&1
tools/profile_run.sh r03/prof_heads16 --heads 16 --no-extra > /dev/null 2>&1
python tools/gemm_bench.py --only E > "$OUT/gemm_fp32.txt" 2>&1
python tools/gemm_bench.py --only E --no-p8 > "$OUT/gemm_fp32_no_p8.txt" 2>&1
python tools/p8_check.py --no-check > "$OUT/gemm_bf16_half.txt" 2>&1
python tools/p8_check.py --no-check --ablate --rows 98304 --only kproj > "$OUT/gemm_p8_ablation.txt" 2>&1
python tools/p8_check.py --no-check --ablate --rows 98304 --only kproj --cold 6 --iters 300 > "$OUT/gemm_p8_ablation_cold.txt" 2>&1
python tools/p8_check.py --no-check --rows 98304 --cold 6 --iters 300 > "$OUT/gemm_bf16_half_cold.txt" 2>&1
python tools/p8_check.py --no-check --ablate --only gather > "$OUT/gemm_p8_gather_ablation.txt" 2>&1
python tools/p8_check.py --no-check --ablate --only gather --structured >> "$OUT/gemm_p8_gather_ablation.txt" 2>&1
python tools/gemm_bench.py --prec 3 --only E --fmt 5 > "$OUT/gemm_bf16x3.txt" 2>&1
python tools/gemm_bench.py --prec 3 --only E --fmt 4101 > "$OUT/gemm_bf16x3_no_p8.txt" 2>&1
python tools/latency_probe.py > "$OUT/latency_fp32.txt" 2>&1
python tools/latency_probe.py --gemm-precision bf16x3 > "$OUT/latency_bf16x3.txt" 2>&1
python tools/latency_probe.py --gemm-precision bf16_mixed > "$OUT/latency_bf16_mixed.txt" 2>&1
for hd in 4 16; do for on in 1 0; do   # NUM_HEADS 4 / 16: the MFMA gate (and the head-dim flash kernel) against the VALU gate of round 2
  python bench.py --no-cpu --no-extra --steps 5 --warmup 2 --heads $hd --debug-option gate_heads_mfma=$on 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('NUM_HEADS=$hd gate_heads_mfma=$on: %.1f scenes/s, %.2f ms/step; gate %.2f TF (%.0f %% of the step), edge attention %.1f TF (%.0f %%)' % (d['value'], d['ms_per_step'], r['class_tflops']['edge_gate'], 100 * r['time_share']['edge_gate'], r['class_tflops']['flash_attn_f32'], 100 * r['time_share']['flash_attn_f32']))"
done; done > "$OUT/heads.txt" 2>&1
for m in bf16_mixed bf16x3; do for hd in 4 16; do
  python bench.py --no-cpu --no-extra --steps 5 --warmup 2 --heads $hd --gemm-precision $m 2>/dev/null | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('NUM_HEADS=$hd $m: %.1f scenes/s, %.2f ms/step; edge attention %.1f TF (%.0f %% of the step), gate %.1f TF (%.0f %%)' % (d['value'], d['ms_per_step'], r['class_tflops']['flash_attn_f32'], 100 * r['time_share']['flash_attn_f32'], r['class_tflops']['edge_gate'], 100 * r['time_share']['edge_gate']))"
done; done >> "$OUT/heads.txt" 2>&1
python tools/eval_synth.py > "$OUT/eval_synth.txt" 2>&1
python tools/switch_scan.py > "$OUT/switch_scan.txt" 2>&1
python tools/fuzz_forward.py --iters 120 > "$OUT/fuzz_forward.txt" 2>&1
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > "$OUT/tests_gpu.log"
du -sh "$OUT"; ls "$OUT"
