#!/bin/bash
set -o pipefail
O=gpurun_out/s10
mkdir -p $O
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_gpu_primitives.py tests/test_gpu_regression.py > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python tools/trace_bench.py 2>&1 | grep -E "grad_trace|gram" || exit 1
timeout -k 10 200 python tools/gram_bench.py 2>&1 | grep -E "gram" || exit 1
