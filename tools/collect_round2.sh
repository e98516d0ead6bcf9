#!/bin/bash
# Everything DESIGN.md quotes for round 2, in one GPU session; outputs under gpurun_out/r02/ (summaries are copied to profiles/).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r02
mkdir -p "$OUT"
cd "$ROOT"
python bench.py > "$OUT/bench_fp32.json" 2> "$OUT/bench_fp32.err"
python bench.py --gemm-precision bf16x3 > "$OUT/bench_cfg3_bf16x3.json" 2> "$OUT/bench_cfg3_bf16x3.err"
python bench.py --gemm-precision bf16_mixed > "$OUT/bench_cfg3_bf16_mixed.json" 2> "$OUT/bench_cfg3_bf16_mixed.err"
python bench.py --no-cpu --gemm-precision bf16 > "$OUT/bench_cfg3_bf16.json" 2> "$OUT/bench_cfg3_bf16.err"
for m in fp32 bf16x3; do
  python bench.py --scenes 1 --objects 200 --points 1024 --steps 10 --warmup 2 --no-cpu --gemm-precision $m > "$OUT/bench_cfg5_$m.json" 2> "$OUT/bench_cfg5_$m.err"
done
tools/profile_run.sh r02/prof_fp32 > /dev/null 2>&1
tools/profile_run.sh r02/prof_cfg3 --gemm-precision bf16x3 > /dev/null 2>&1
tools/profile_run.sh r02/prof_cfg3_mixed --gemm-precision bf16_mixed > /dev/null 2>&1
python tools/latency_probe.py > "$OUT/latency_fp32.txt" 2>&1
python tools/latency_probe.py --gemm-precision bf16x3 > "$OUT/latency_bf16x3.txt" 2>&1
tools/single_scene_trace.sh r02/single_fp32 --single-only > /dev/null 2>&1
tools/single_scene_trace.sh r02/single_bf16x3 --single-only --gemm-precision bf16x3 > /dev/null 2>&1
python tools/gemm_bench.py --only E > "$OUT/gemm_fp32.txt" 2>&1
python tools/gemm_bench.py --prec 3 --only E --fmt 5 --prefetch 0,6 > "$OUT/gemm_bf16x3.txt" 2>&1
python tools/gemm_bench.py --prec 3 --only E --fmt 21 --prefetch 6 > "$OUT/gemm_bf16x3_noring.txt" 2>&1
python tools/gemm_bench.py --prec 3 --only E --no-dma > "$OUT/gemm_bf16x3_vgpr.txt" 2>&1
python tools/gemm_bench.py --prec 1 --only E --fmt 5 > "$OUT/gemm_bf16.txt" 2>&1
python tools/gemm_bench.py --prec 1 --only E --fmt 37 > "$OUT/gemm_bf16_half.txt" 2>&1
{ echo "ring kernel, kv launch (E x 1024 x 512): full | no operand loads after the first slices | no MFMAs | neither"
  for f in 5 261 517 773; do python tools/gemm_bench.py --prec 3 --only "kv E" --fmt $f 2>&1 | grep " E x"; done
  for f in 37 293 549 805; do python tools/gemm_bench.py --prec 1 --only "kv E" --fmt $f 2>&1 | grep " E x"; done
  echo "128 x 256 tiles instead of 256 x 128 (fmt bit 7)"
  python tools/gemm_bench.py --prec 3 --only "kv E" --fmt 133 2>&1 | grep " E x"
  python tools/gemm_bench.py --prec 1 --only "kv E" --fmt 165 2>&1 | grep " E x"; } > "$OUT/gemm_ablation.txt" 2>&1
python tools/gemm_clock_probe.py > "$OUT/gemm_clock_fp32.txt" 2>&1
python tools/gemm_clock_probe.py --prec 3 > "$OUT/gemm_clock_bf16x3.txt" 2>&1
python tools/gemm_clock_probe.py --prec 1 > "$OUT/gemm_clock_bf16.txt" 2>&1
tools/forward_timeline.sh r02 40 > /dev/null 2>&1
tools/forward_timeline.sh r02 40 bf16x3 > /dev/null 2>&1 && mv "$OUT/timeline_40.txt" "$OUT/timeline_40_bf16x3.txt"; tools/forward_timeline.sh r02 40 > /dev/null 2>&1
# (tools/graph_replay_probe.py and the hipGraph replay path it measured -- vlsat_forward_graph -- were removed in round 5: slower than eager)
tools/bin/l2_fill_probe > "$OUT/l2_fill.txt" 2>&1
tools/bin/lds_bw_probe > "$OUT/lds_bw.txt" 2>&1
tools/bin/tr_read_probe > "$OUT/tr_read.txt" 2>&1
python tools/eval_synth.py > "$OUT/eval_synth.txt" 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --hip-runtime-trace --stats -d "$OUT/api" -o api --output-format csv -- python "$ROOT/tools/api_trace_forward.py" > "$OUT/api_trace.txt" 2> "$OUT/api_trace.log" )
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/api/**/*hip_api_stats.csv", recursive=True) + glob.glob(out + "/api/**/*_stats.csv", recursive=True):
    rows = list(csv.reader(open(f)))
    open(out + "/api_stats_" + f.split("/")[-1], "w").write("\n".join(",".join(r) for r in rows[:40]) + "\n")
PY
rm -rf "$OUT/api"
du -sh "$OUT"
ls "$OUT"
