#!/bin/bash
# round 6, fourth GPU session: K-tile rotation (step A/B per mode, FETCH_SIZE with / without), asm V reads of the edge attention
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_fourth
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_hip_round6.py -x -q > "$OUT/tests.txt" 2>&1; tail -5 "$OUT/tests.txt"
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "p8 or flash" >> "$OUT/tests.txt" 2>&1; tail -3 "$OUT/tests.txt"
: > "$OUT/step_ab.txt"
one() {  # mode, steps, extra args...
  local mode=$1 steps=$2; shift 2
  python bench.py --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'
}
for rep in 1 2 3; do
  for mode in bf16x3 bf16x3_attn1 fp32; do
    steps=30; [ $mode = fp32 ] && steps=15
    for opt in "gemm_k_rot=0" "gemm_k_rot=1"; do
      echo "$mode $opt rep=$rep $(one $mode $steps --debug-option $opt)" >> "$OUT/step_ab.txt"
    done
  done
  for opt in "flash_asmv=0" "flash_asmv=1"; do
    echo "bf16_mixed $opt rep=$rep $(one bf16_mixed 40 --debug-option $opt)" >> "$OUT/step_ab.txt"
    echo "cfg5 bf16_mixed $opt rep=$rep $(one bf16_mixed 6 --scenes 1 --objects 200 --points 1024 --debug-option $opt)" >> "$OUT/step_ab.txt"
  done
done
cat "$OUT/step_ab.txt"
cd /tmp && export TMPDIR=/tmp
for r in 0 1; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/pmc_${ctr}_rot$r" -o p --output-format csv -- python "$ROOT/bench.py" --gemm-precision bf16_mixed --steps 3 --warmup 1 --no-cpu --no-profile --no-extra --debug-option dual_stream=0 --debug-option gemm_k_rot=$r > "$OUT/pmc_${ctr}_rot$r.log" 2>&1
  done
  python - "$OUT" $r <<'PY'
import sys, glob, csv, collections
out, r = sys.argv[1], sys.argv[2]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_{ctr}_rot{r}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no csv for", ctr, r); continue
    tot = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") != ctr: continue
        k = row["Kernel_Name"].split("(")[0][-60:]
        tot[k][0] += 1; tot[k][1] += float(row["Counter_Value"])
    allb = sum(v[1] for v in tot.values())
    print(f"k_rot={r} {ctr}: total {allb:.4g} (raw units) over {sum(v[0] for v in tot.values())} launches")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"   {k:60s} n={v[0]:4d} per launch {v[1]/v[0]:.4g}")
PY
done 2>&1 | tee "$OUT/pmc_rot.txt"
rm -rf "$OUT"/pmc_*_rot?/
