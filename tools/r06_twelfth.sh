#!/bin/bash
# round 6: the one-scene loop at 2 / 4 / 6 in flight under the split-K bound and the fused aggregation, interleaved repetitions
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_twelfth
mkdir -p "$OUT"; cd "$ROOT"
for rep in 1 2 3; do
  for cfg in "gemm_splitk_max_tiles=0 gate_fuse_agg=1" "gemm_splitk_max_tiles=64 gate_fuse_agg=1" "gemm_splitk_max_tiles=0 gate_fuse_agg=2" "gemm_splitk_max_tiles=64 gate_fuse_agg=2"; do
    set -- $cfg
    echo "== rep $rep: $cfg"
    python tools/val_loop_probe.py --workers 2,4,6 --merge "" --debug-option $1 --debug-option $2 2>&1 | grep -E "in flight"
  done
done | tee "$OUT/loop_ab.txt"
