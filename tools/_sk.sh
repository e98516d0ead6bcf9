mkdir -p gpurun_out/r2o
python tools/gemm_bench.py --only E > gpurun_out/r2o/gemm_fp32.txt 2>&1
python tools/gemm_bench.py --prec 1 --only "E x" --fmt 37 > gpurun_out/r2o/gemm_half.txt 2>&1
python tools/gemm_bench.py --prec 3 --only "E x" --fmt 5 > gpurun_out/r2o/gemm_x3.txt 2>&1
python bench.py --no-cpu > gpurun_out/r2o/bench.json 2>/dev/null
python bench.py --no-cpu --gemm-precision bf16x3 > gpurun_out/r2o/bench_x3.json 2>/dev/null
python bench.py --no-cpu --gemm-precision bf16_mixed > gpurun_out/r2o/bench_mixed.json 2>/dev/null
(timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k gemm --timeout 600 2>&1 | tail -5) > gpurun_out/r2o/tests.log
cat gpurun_out/r2o/tests.log; grep -h "out-proj\|nn_edge.0\|kv E" gpurun_out/r2o/gemm_*.txt
python - <<'PY'
import json
for f in ("bench", "bench_x3", "bench_mixed"):
    d = json.loads([l for l in open(f"gpurun_out/r2o/{f}.json") if l.startswith("{")][0])
    print(f, d["value"], d["roofline"]["class_tflops"])
PY
