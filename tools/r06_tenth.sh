#!/bin/bash
# round 6: which launches should take the split-K kernel (tile bound), and the fused max aggregation in exact fp32 -- batch step AND one-scene latency / loop
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_tenth
mkdir -p "$OUT"; cd "$ROOT"
one() { local mode=$1 steps=$2; shift 2; python bench.py --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2; do
  for t in 0 128 64 32; do
    echo "bf16_mixed gemm_splitk_max_tiles=$t rep=$rep $(one bf16_mixed 40 --debug-option gemm_splitk_max_tiles=$t)" >> "$OUT/ab.txt"
    echo "fp32 gemm_splitk_max_tiles=$t rep=$rep $(one fp32 15 --debug-option gemm_splitk_max_tiles=$t)" >> "$OUT/ab.txt"
    echo "bf16x3 gemm_splitk_max_tiles=$t rep=$rep $(one bf16x3 25 --debug-option gemm_splitk_max_tiles=$t)" >> "$OUT/ab.txt"
  done
  echo "fp32 gemm_splitk_max_tiles=128+gate_fuse_agg=2 rep=$rep $(one fp32 15 --debug-option gemm_splitk_max_tiles=128 --debug-option gate_fuse_agg=2)" >> "$OUT/ab.txt"
done
cat "$OUT/ab.txt"
for t in 0 128 64 32; do
  echo "== one scene per call, gemm_splitk_max_tiles=$t"
  python tools/latency_probe.py --single-only --debug-option gemm_splitk_max_tiles=$t 2>&1 | grep -E "same graphs|objects \("
done | tee "$OUT/latency.txt"
for o in "gate_fuse_agg=1" "gate_fuse_agg=2"; do
  echo "== one scene per call, $o"
  python tools/latency_probe.py --single-only --debug-option $o 2>&1 | grep -E "same graphs|objects \("
done | tee -a "$OUT/latency.txt"
