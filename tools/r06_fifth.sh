#!/bin/bash
# round 6, fifth GPU session: full GPU suite on the new defaults, the one-scene loop after the ranking change, default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_fifth
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests -q -m gpu -x > "$OUT/tests_gpu.txt" 2>&1; tail -6 "$OUT/tests_gpu.txt"
python tools/val_loop_probe.py > "$OUT/val_loop_fp32.txt" 2>&1; cat "$OUT/val_loop_fp32.txt"
python tools/val_loop_probe.py --gemm-precision bf16_mixed > "$OUT/val_loop_bf16_mixed.txt" 2>&1; cat "$OUT/val_loop_bf16_mixed.txt"
python tools/latency_probe.py > "$OUT/latency_fp32.txt" 2>&1; head -12 "$OUT/latency_fp32.txt"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print("value", j["value"], j["ms_per_step"], "roofline", j["roofline"]["frac"], j["roofline"]["traffic"], "eval", j["evaluation"]["scenes_per_s_per_gpu"], j["evaluation"]["reference_compatible_rank_lists"]["scenes_per_s_per_gpu"], "cpu", j["cpu_baseline"]["value"], j["max_abs_err_vs_cpu_oracle"])
for x in j["extra_configs"] or []: print(x["workload"][-34:], x["value"], x["ms_per_step"], x["max_abs_err_vs_cpu_oracle"], (x["roofline"] or {}).get("frac"), (x["roofline"] or {}).get("peak_note"))
PY
