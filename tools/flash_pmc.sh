#!/bin/bash
# SQ counter passes over the cfg 5 scene (the edge attention is 75 % of its bf16_mixed step): tools/flash_pmc.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -f "$OUT/counters.txt" ] || rocprofv3 -L > "$OUT/counters.txt" 2>&1
pass() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc_$name" -o p --output-format csv -- python "$ROOT/bench.py" --scenes 1 --objects 200 --points 1024 --steps 3 --warmup 1 --no-cpu --no-profile --no-extra --gemm-precision bf16_mixed "${ARGS[@]}" > "$OUT/pmc_$name.log" 2>&1
  python "$ROOT/tools/pmc_raw.py" "$OUT/pmc_$name" flash_attn_bf16 > "$OUT/flash_$name.txt" 2>&1; rm -rf "$OUT/pmc_$name"; }
ARGS=("$@")
pass a SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass b SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cat "$OUT"/flash_a.txt "$OUT"/flash_b.txt
