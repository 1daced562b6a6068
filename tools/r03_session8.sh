#!/bin/bash
set -o pipefail
O=gpurun_out/s8
mkdir -p $O
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_gpu_ep.py tests/test_gpu_fullsize.py -k "ep" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 200 python tools/ep_bench.py 2>&1 | grep -v amdgpu
timeout -k 10 100 python tools/ep_fit_time.py 2>&1 | grep -v amdgpu
