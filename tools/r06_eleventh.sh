#!/bin/bash
# round 6: on the library with the split-K bound -- GPU suite, fused aggregation in fp32 (batch + loop), size bound of the paired schedule
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_eleventh
mkdir -p "$OUT"; cd "$ROOT"
timeout 1500 python -m pytest tests -q -m gpu -x > "$OUT/tests.txt" 2>&1; tail -3 "$OUT/tests.txt"
one() { local mode=$1 steps=$2; shift 2; python bench.py --gemm-precision $mode --steps $steps --warmup 4 --no-cpu --no-extra --no-profile "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"])'; }
: > "$OUT/ab.txt"
for rep in 1 2 3; do
  for o in "gate_fuse_agg=1" "gate_fuse_agg=2"; do echo "fp32 $o rep=$rep $(one fp32 15 --debug-option $o)" >> "$OUT/ab.txt"; done
done
cat "$OUT/ab.txt"
for o in "pair_max_edges=4096" "pair_max_edges=2048" "pair_max_edges=8192" "gate_fuse_agg=2" "pair_twins=0"; do
  echo "== loop, fp32 mix, $o"
  python tools/val_loop_probe.py --workers 1,4 --merge "" --debug-option $o 2>&1 | grep -E "in flight"
done | tee "$OUT/loop.txt"
