#!/bin/bash
# round 6: longer soak of the paths this round touched (paired schedule under replicas, K-tile rotation, new ranking kernels)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_soak
mkdir -p "$OUT"; cd "$ROOT"
{
for mode in fp32 bf16_mixed bf16x3_attn1; do
  echo "== replica_race_probe $mode (400 one-scene graphs, 5 replicas on 5 threads, 6 passes; paired schedule on)"
  python tools/replica_race_probe.py --scenes 400 --passes 6 --gemm-precision $mode 2>&1 | grep -v amdgpu | tail -3
done
echo "== fuzz_forward 500 random configurations / graphs against the CPU oracle"
python tools/fuzz_forward.py --iters 500 --seed 6 2>&1 | grep -v amdgpu | tail -2
echo "== soak_forward (batched + one-scene forwards must reproduce the first run bit for bit)"
python tools/soak_forward.py --iters 600 2>&1 | grep -v amdgpu | tail -4
echo "== evaluation loop summaries, 3 repetitions of val_loop_probe (summaries asserted equal inside)"
for i in 1 2 3; do python tools/val_loop_probe.py --scenes 60 --workers 1,3,5 --merge 4 2>&1 | grep -v amdgpu | grep -E "in flight|ONE call" | head -5; done
} > "$OUT/long_soak.txt" 2>&1
cat "$OUT/long_soak.txt"
