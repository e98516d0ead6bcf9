#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_17
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt" -o t -- python "$ROOT/bench.py" --gemm-precision bf16_mixed --steps 3 --warmup 2 --no-cpu --no-extra --no-profile > "$OUT/bench.json" 2> "$OUT/kt.log"
python "$ROOT/tools/lane_timeline.py" "$OUT"/kt/t_results.db "$OUT/lanes_bf16_mixed.txt"
python - "$OUT"/kt/t_results.db <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); print([r[1] for r in c.execute("pragma table_info(kernels)")])
PY
rm -rf "$OUT/kt"
