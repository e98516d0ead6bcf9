#!/bin/bash
# Copies what DESIGN.md quotes from gpurun_out/r02/ (scratch, written by tools/collect_round2.sh on the GPU box) to profiles/ (tracked).
set -eu
cd "$(dirname "$0")/.."
S=gpurun_out/r02; D=profiles; P=$D/r02_probes
mkdir -p $P
j() { grep '^{' "$1" | tail -1 > "$2"; }
j $S/bench_fp32.json $D/r02_bench_fp32.json
j $S/bench_cfg3_bf16x3.json $D/r02_cfg3_bf16x3_bench.json
j $S/bench_cfg3_bf16_mixed.json $D/r02_cfg3_bf16_mixed_bench.json
j $S/bench_cfg3_bf16.json $D/r02_cfg3_bf16_bench.json
j $S/bench_cfg5_fp32.json $D/r02_cfg5_fp32_bench.json
j $S/bench_cfg5_bf16x3.json $D/r02_cfg5_bf16x3_bench.json
cp $S/prof_fp32/kernel_stats.md $D/r02_bench_kernel_stats.md;        cp $S/prof_fp32/pmc.md $D/r02_bench_pmc.md
cp $S/prof_fp32/pmc.json $D/r02_bench_pmc.json; cp $S/prof_cfg3/pmc.json $D/r02_cfg3_bf16x3_pmc.json; cp $S/prof_cfg3_mixed/pmc.json $D/r02_cfg3_bf16_mixed_pmc.json
cp $S/prof_cfg3/kernel_stats.md $D/r02_cfg3_bf16x3_kernel_stats.md;  cp $S/prof_cfg3/pmc.md $D/r02_cfg3_bf16x3_pmc.md
cp $S/prof_cfg3_mixed/kernel_stats.md $D/r02_cfg3_bf16_mixed_kernel_stats.md; cp $S/prof_cfg3_mixed/pmc.md $D/r02_cfg3_bf16_mixed_pmc.md
for m in fp32 bf16x3; do
  { cat $S/single_$m/trace_summary.txt; echo; grep -v amdgpu.ids $S/single_$m/latency.txt; } > $D/r02_single_scene_$m.txt
  grep -v amdgpu.ids $S/latency_$m.txt > $D/r02_latency_$m.txt
done
cp $S/timeline_40.txt $D/r02_forward_timeline_40.txt
cp $S/timeline_40_bf16x3.txt $D/r02_forward_timeline_40_bf16x3.txt
{ grep -v amdgpu.ids $S/api_trace.txt; echo; echo "rocprofv3 --hip-runtime-trace --stats, HIP API calls by total time:"; cat $S/api_stats_api_hip_api_stats.csv; } > $D/r02_hip_api_trace.txt
for f in gemm_fp32 gemm_bf16x3 gemm_bf16x3_noring gemm_bf16x3_vgpr gemm_bf16 gemm_bf16_half gemm_ablation gemm_clock_fp32 gemm_clock_bf16x3 gemm_clock_bf16 graph_replay l2_fill lds_bw tr_read eval_synth; do
  grep -v amdgpu.ids $S/$f.txt > $P/$f.txt
done
cp $S/tests_gpu.log $D/r02_tests_gpu.txt
ls $D $P
