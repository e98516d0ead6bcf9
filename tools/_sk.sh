python tools/gemm_bench.py --prec 1 --only "E x" --fmt 37 2>&1 | grep " E x" | grep -v "conv\|fc3"
(timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "ring or half_row" --timeout 600 2>&1 | tail -3)
