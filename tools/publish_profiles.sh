#!/bin/bash
# Copies what DESIGN.md quotes from gpurun_out/<round>/ (scratch, written by tools/collect_round<N>.sh on the GPU box) to
# profiles/ (tracked):   tools/publish_profiles.sh r03
set -u
cd "$(dirname "$0")/.."
R=${1:-r03}
S=gpurun_out/$R; D=profiles; P=$D/${R}_probes
mkdir -p $P
j() { grep '^{' "$1" | tail -1 > "$2"; }
j $S/bench_fp32.json $D/${R}_bench_fp32.json
j $S/bench_cfg3_bf16x3.json $D/${R}_cfg3_bf16x3_bench.json
j $S/bench_cfg3_bf16_mixed.json $D/${R}_cfg3_bf16_mixed_bench.json
cp $S/prof_fp32/kernel_stats.md $D/${R}_bench_kernel_stats.md;        cp $S/prof_fp32/pmc.md $D/${R}_bench_pmc.md
for t in fp32:bench cfg3:cfg3_bf16x3 cfg3_mixed:cfg3_bf16_mixed cfg3_attn1:cfg3_bf16x3_attn1 cfg3_f16:cfg3_fp16_mixed cfg5_fp32:cfg5_fp32 cfg5_mixed:cfg5_bf16_mixed; do   # (round 4 on) serialised traces with the roofline footer; cfg 5 PMC
  a=${t%%:*}; b=${t##*:}
  [ -f $S/prof_$a/kernel_stats_serial.md ] && cp $S/prof_$a/kernel_stats_serial.md $D/${R}_${b}_kernel_stats_serial.md
  case $a in cfg5_*|cfg3_attn1|cfg3_f16) [ -f $S/prof_$a/pmc.json ] && { cp $S/prof_$a/pmc.json $D/${R}_${b}_pmc.json; cp $S/prof_$a/pmc.md $D/${R}_${b}_pmc.md; cp $S/prof_$a/kernel_stats.md $D/${R}_${b}_kernel_stats.md; };; esac
done
cp $S/prof_fp32/pmc.json $D/${R}_bench_pmc.json; cp $S/prof_cfg3/pmc.json $D/${R}_cfg3_bf16x3_pmc.json; cp $S/prof_cfg3_mixed/pmc.json $D/${R}_cfg3_bf16_mixed_pmc.json
cp $S/prof_cfg3/kernel_stats.md $D/${R}_cfg3_bf16x3_kernel_stats.md;  cp $S/prof_cfg3/pmc.md $D/${R}_cfg3_bf16x3_pmc.md
cp $S/prof_cfg3_mixed/kernel_stats.md $D/${R}_cfg3_bf16_mixed_kernel_stats.md; cp $S/prof_cfg3_mixed/pmc.md $D/${R}_cfg3_bf16_mixed_pmc.md
for hd in 4 16; do [ -d $S/prof_heads$hd ] && { cp $S/prof_heads$hd/kernel_stats.md $D/${R}_heads${hd}_kernel_stats.md; cp $S/prof_heads$hd/pmc.md $D/${R}_heads${hd}_pmc.md; }; done
for m in fp32 bf16x3 bf16_mixed; do [ -f $S/latency_$m.txt ] && grep -v amdgpu.ids $S/latency_$m.txt > $D/${R}_latency_$m.txt; done
for f in gemm_fp32 gemm_fp32_no_p8 gemm_bf16_half gemm_bf16_half_cold gemm_p8_ablation gemm_p8_ablation_cold gemm_p8_gather_ablation heads switch_scan fuzz_forward gemm_bf16x3 gemm_bf16x3_no_p8 eval_synth \
         gemm_tile_sweep val_loop_fp32 val_loop_bf16_mixed val_loop_fp32_n40 stress_scan soak_forward replica_race_fp32 replica_race_bf16_mixed replica_race_fp16_mixed latency_fp32_unpaired \
         timeline_20 timeline_20_unpaired timeline_40 timeline_40_unpaired metrics_bench; do
  [ -f $S/$f.txt ] && grep -v amdgpu.ids $S/$f.txt > $P/$f.txt
done
for l in launches_fp32 launches_bf16_mixed; do [ -f $S/$l/launches.txt ] && cp $S/$l/launches.txt $P/$l.txt; done
cp $S/tests_gpu.log $D/${R}_tests_gpu.txt
[ -f $S/eval_kernel_stats.md ] && cp $S/eval_kernel_stats.md $D/${R}_eval_kernel_stats.md
[ -f $S/eval_pmc.md ] && cp $S/eval_pmc.md $D/${R}_eval_pmc.md
ls $D $P
# stamp the published summaries with the commit they were published from (the GPU box has no .git; the digests of sources and
# library in `collected_on` were taken there)
python - "$R" <<'PY'
import glob, json, subprocess, sys
head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
for f in glob.glob(f"profiles/{sys.argv[1]}_*_pmc.json") + glob.glob(f"profiles/{sys.argv[1]}_bench_pmc.json"):
    j = json.load(open(f))
    j.setdefault("collected_on", {})["published_from_git_head"] = head
    json.dump(j, open(f, "w"), indent=1)
PY
