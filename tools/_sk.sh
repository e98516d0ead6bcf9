mkdir -p gpurun_out/r2n
(timeout 900 python -m pytest tests/test_hip_forward.py tests/test_hip_round2.py -m gpu -q -x --timeout 600 2>&1 | tail -5) > gpurun_out/r2n/tests2.log
python tools/latency_probe.py > gpurun_out/r2n/lat.txt 2>&1
python tools/latency_probe.py --debug-option gemm_splitk=0 --single-only > gpurun_out/r2n/lat_nosk.txt 2>&1
python tools/latency_probe.py --debug-option node_attn_split=0 --single-only > gpurun_out/r2n/lat_nonas.txt 2>&1
python tools/latency_probe.py --gemm-precision bf16x3 > gpurun_out/r2n/lat_x3.txt 2>&1
python bench.py --no-cpu > gpurun_out/r2n/bench.json 2>/dev/null
python bench.py --no-cpu --debug-option node_attn_split=0 > gpurun_out/r2n/bench_nonas.json 2>/dev/null
tools/forward_timeline.sh r2n 40 > /dev/null 2>&1
cat gpurun_out/r2n/tests2.log; for f in lat lat_nosk lat_nonas lat_x3; do echo $f; grep -E "new graph|same graphs|back to back|ONE call" gpurun_out/r2n/$f.txt; done
python - <<'PY'
import json
for f in ("bench", "bench_nonas"):
    d = json.loads([l for l in open(f"gpurun_out/r2n/{f}.json") if l.startswith("{")][0])
    print(f, d["value"], d["roofline"]["time_share"])
PY
