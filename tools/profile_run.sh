#!/bin/bash
# rocprofv3 evidence for one bench.py configuration, written under gpurun_out/<tag>/ (copy what you quote to profiles/):
#   tools/profile_run.sh <tag> [bench.py arguments ...]
#   1. kernel trace + stats of `bench.py --steps 5 --warmup 2 --no-cpu <args>`           -> <tag>/kernel_stats.md, bench.json
#   2. PMC pass, SQ counters  (matrix-pipe busy, LDS conflicts, wait / issue-stall / active cycles)
#   3. PMC pass, FETCH_SIZE   4. PMC pass, WRITE_SIZE   5. PMC pass, L2 hit / miss / requests
# Counter passes run `bench.py --steps 3 --warmup 1 --no-cpu --no-profile <args>`; each in its own rocprofv3 run with
# --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o bench -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu "$@" > "$OUT/bench.json" 2> "$OUT/kt.log"
python "$ROOT/tools/rocprof_summary.py" "$OUT"/kt/bench_results.db "$OUT/kernel_stats.md" "$OUT/bench.json" > /dev/null 2>> "$OUT/kt.log"
# 1b. the same trace SERIALISED (one stream): kernel durations are then the kernels' own, and the footer of the table
#     recomputes the bench line's roofline from the trace alone                         -> <tag>/kernel_stats_serial.md
rocprofv3 --kernel-trace --stats -d "$OUT/kts" -o bench -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu --debug-option dual_stream=0 "$@" > "$OUT/bench_serial.json" 2> "$OUT/kts.log"
python "$ROOT/tools/rocprof_summary.py" "$OUT"/kts/bench_results.db "$OUT/kernel_stats_serial.md" "$OUT/bench_serial.json" > /dev/null 2>> "$OUT/kts.log"
pass() {  # name, counters...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/pmc_$name" -o p --output-format csv -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu --no-profile "${ARGS[@]}" > "$OUT/pmc_$name.log" 2>&1
}
ARGS=("$@")
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ
python "$ROOT/tools/pmc_summary.py" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc.md" "$OUT/pmc_l2" > /dev/null 2> "$OUT/pmc_summary.err"
# the raw traces are tens of MB; gpurun merges at most 64 MiB back (KEEP_RAW=1 keeps them)
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf "$OUT/kt" "$OUT/kts" "$OUT"/pmc_sq "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_l2
ls "$OUT"
