#!/bin/bash
# round 6: lab switches re-measured under the three-lane schedule (experiments build): gate grid, node-attention split
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_16
mkdir -p "$OUT"; cd "$ROOT"
one() { local mode=$1 steps=$2; shift 2; python bench.py --lib tools/bin/libvlsat_hip_exp.so --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2 3; do
  for g in 0 768 1280 1536 2048; do echo "bf16_mixed gate_grid=$g rep=$rep $(one bf16_mixed 40 --debug-option gate_grid=$g)" >> "$OUT/ab.txt"; done
  for g in 0 512 1024 1536; do echo "fp32 gate_grid=$g rep=$rep $(one fp32 15 --debug-option gate_grid=$g)" >> "$OUT/ab.txt"; done
  for v in 1024 0 4096; do echo "bf16_mixed node_attn_split=$v rep=$rep $(one bf16_mixed 40 --debug-option node_attn_split=$v)" >> "$OUT/ab.txt"; done
done
cat "$OUT/ab.txt"
