#!/bin/bash
# Everything DESIGN.md quotes for round 6, in GPU sessions of one part each; outputs under gpurun_out/r06/ (tools/publish_profiles.sh r06
# copies the summaries to profiles/).  PART=a|b|c.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06
PART=${PART:-abc}
mkdir -p "$OUT"
cd "$ROOT"
if [[ $PART == *a* ]]; then
python bench.py > "$OUT/bench_fp32.json" 2> "$OUT/bench_fp32.err"                       # the driver's command: headline + extra_configs
python bench.py --gemm-precision bf16x3 --no-extra > "$OUT/bench_cfg3_bf16x3.json" 2> "$OUT/bench_cfg3_bf16x3.err"
python bench.py --gemm-precision bf16_mixed --no-extra > "$OUT/bench_cfg3_bf16_mixed.json" 2> "$OUT/bench_cfg3_bf16_mixed.err"
tools/profile_run.sh r06/prof_fp32 --no-extra > /dev/null 2>&1
tools/profile_run.sh r06/prof_cfg3 --gemm-precision bf16x3 --no-extra > /dev/null 2>&1
tools/profile_run.sh r06/prof_cfg3_mixed --gemm-precision bf16_mixed --no-extra > /dev/null 2>&1
tools/profile_run.sh r06/prof_cfg3_attn1 --gemm-precision bf16x3_attn1 --no-extra > /dev/null 2>&1
tools/profile_run.sh r06/prof_cfg3_f16 --gemm-precision fp16_mixed --no-extra > /dev/null 2>&1
fi
if [[ $PART == *b* ]]; then
tools/profile_run.sh r06/prof_cfg5_fp32 --scenes 1 --objects 200 --points 1024 --no-extra > /dev/null 2>&1
tools/profile_run.sh r06/prof_cfg5_mixed --scenes 1 --objects 200 --points 1024 --gemm-precision bf16_mixed --no-extra > /dev/null 2>&1
tools/launch_list.sh r06/launches_fp32 --debug-option dual_stream=0 > /dev/null 2>&1
tools/launch_list.sh r06/launches_bf16_mixed --gemm-precision bf16_mixed --debug-option dual_stream=0 > /dev/null 2>&1
python tools/latency_probe.py > "$OUT/latency_fp32.txt" 2>&1
python tools/latency_probe.py --gemm-precision bf16_mixed > "$OUT/latency_bf16_mixed.txt" 2>&1
python tools/latency_probe.py --debug-option pair_twins=0 --single-only > "$OUT/latency_fp32_unpaired.txt" 2>&1
python tools/val_loop_probe.py > "$OUT/val_loop_fp32.txt" 2>&1
python tools/val_loop_probe.py --gemm-precision bf16_mixed > "$OUT/val_loop_bf16_mixed.txt" 2>&1
python tools/val_loop_probe.py --objects 40 --workers 1,2,4,8 > "$OUT/val_loop_fp32_n40.txt" 2>&1
for n in 20 40; do
  tools/forward_timeline.sh r06 $n fp32 256 "" "" > /dev/null 2>&1
  tools/forward_timeline.sh r06 $n fp32 256 pair_twins=0 _unpaired > /dev/null 2>&1
done
tools/forward_timeline.sh r06 40 bf16_mixed 256 "" _bf16_mixed > /dev/null 2>&1
# the step after the path: kernel stats + counters of the ranking kernels
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d "$OUT/kt_eval" -o ev -- python "$ROOT/tools/metrics_bench.py" > "$OUT/metrics_bench.txt" 2> "$OUT/kt_eval.log"
  python "$ROOT/tools/rocprof_summary.py" "$OUT"/kt_eval/ev_results.db "$OUT/eval_kernel_stats.md" > /dev/null 2>> "$OUT/kt_eval.log"
  for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
    set -- $pass; name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc_eval_$name" -o p --output-format csv -- python "$ROOT/tools/metrics_bench.py" > "$OUT/pmc_eval_$name.log" 2>&1
  done
  python - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
want = ("tri_rank_kernel", "rel_rank_kernel", "obj_rank_kernel", "sort_probs_kernel", "softmax_rows_kernel", "eval_counts_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for name in ("sq", "fetch", "write"):
    for f in glob.glob(f"{out}/pmc_eval_{name}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = next((w for w in want if w in row["Kernel_Name"]), None)
            if not k: continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row.get("Dispatch_Id"))
            if name == "sq" and key not in seen and row["Counter_Name"] == "SQ_WAVE_CYCLES":
                seen.add(key); n[k] += 1
lines = ["| kernel | launches | wave cycles: waiting / issue-stalled / issuing % | LDS bank-conflict % | HBM read KB/launch (2 x FETCH_SIZE) | HBM write KB/launch |", "|---|---|---|---|---|---|"]
for k in want:
    a = acc[k]; c = max(n[k], 1)
    wc = max(a.get("SQ_WAVE_CYCLES", 0), 1)
    lines.append(f"| `{k}` | {n[k]} | {100*a.get('SQ_WAIT_ANY',0)/wc:.0f} / {100*a.get('SQ_WAIT_INST_ANY',0)/wc:.0f} / {100*a.get('SQ_ACTIVE_INST_ANY',0)/wc:.0f} | "
                 f"{100*a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',0),1):.1f} | {2*a.get('FETCH_SIZE',0)/c:.0f} | {a.get('WRITE_SIZE',0)/c:.0f} |")
open(f"{out}/eval_pmc.md", "w").write("\n".join(lines) + "\n(64-scene batch of tools/metrics_bench.py: N = 2560 nodes, E = 99 840 edges, C = 160, R = 26; FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)\n")
print("\n".join(lines))
PY
  rm -rf "$OUT"/kt_eval "$OUT"/pmc_eval_sq "$OUT"/pmc_eval_fetch "$OUT"/pmc_eval_write )
fi
if [[ $PART == *c* ]]; then
python tools/eval_synth.py > "$OUT/eval_synth.txt" 2>&1
python tools/stress_scan.py > "$OUT/stress_scan.txt" 2>&1
python tools/fuzz_forward.py --iters 120 > "$OUT/fuzz_forward.txt" 2>&1
python tools/soak_forward.py > "$OUT/soak_forward.txt" 2>&1
python tools/replica_race_probe.py --scenes 120 --passes 3 > "$OUT/replica_race_fp32.txt" 2>&1
python tools/replica_race_probe.py --scenes 120 --passes 3 --gemm-precision bf16_mixed > "$OUT/replica_race_bf16_mixed.txt" 2>&1
python tools/replica_race_probe.py --scenes 120 --passes 2 --gemm-precision fp16_mixed > "$OUT/replica_race_fp16_mixed.txt" 2>&1
python -m pytest tests -q -m gpu -rf 2>&1 | grep -E "^FAILED|passed|failed|error" | tail -12 > "$OUT/tests_gpu.log"
fi
du -sh "$OUT"; ls "$OUT"
