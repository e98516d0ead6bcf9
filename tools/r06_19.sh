#!/bin/bash
# cross-half reductions by v_permlane32_swap instead of ds_bpermute (common.h half_max / half_sum): whole GPU suite, then interleaved
# A/B of the library against the one before the change (cvpr2023-vlsat_amd/libvlsat_hip_base.so)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_19
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > "$OUT/tests_gpu.log"
tail -3 "$OUT/tests_gpu.log"
{
tools/ab_lib.sh cvpr2023-vlsat_amd/libvlsat_hip_base.so 2 "bf16_mixed fp32 bf16x3"
tools/ab_lib.sh cvpr2023-vlsat_amd/libvlsat_hip_base.so 2 "bf16_mixed fp32" --scenes 1 --objects 200 --points 1024
} 2>&1 | tee "$OUT/ab_permlane_swap.txt"
