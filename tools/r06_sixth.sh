#!/bin/bash
# round 6, sixth GPU session: the paired schedule of one-scene plans -- parity, then latency and loop throughput with / without
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_sixth
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_hip_round6.py -x -q > "$OUT/tests.txt" 2>&1; tail -15 "$OUT/tests.txt"
timeout 1500 python -m pytest tests/test_hip_forward.py tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_kernels.py -x -q >> "$OUT/tests.txt" 2>&1; tail -4 "$OUT/tests.txt"
for pt in 1 0; do
  python tools/latency_probe.py --debug-option pair_twins=$pt > "$OUT/latency_fp32_pair$pt.txt" 2>&1; grep -E "same graphs|objects \(|whole forward" "$OUT/latency_fp32_pair$pt.txt"
done
python tools/val_loop_probe.py --workers 1,2,4,6 --merge "" > "$OUT/val_loop_fp32.txt" 2>&1; cat "$OUT/val_loop_fp32.txt"
python tools/val_loop_probe.py --workers 1,2,4,6 --merge "" --gemm-precision bf16_mixed > "$OUT/val_loop_bf16_mixed.txt" 2>&1; cat "$OUT/val_loop_bf16_mixed.txt"
