#!/bin/bash
set -o pipefail
O=gpurun_out/s3
mkdir -p $O
timeout -k 10 600 python -u -m pytest -x -q --timeout 200 --timeout-method thread -m gpu tests/test_gpu_primitives.py -k "grad_trace or gram" tests/test_gpu_regression.py > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
echo "== trace kernel: built library"; timeout -k 10 200 python tools/trace_bench.py 2>&1 | grep -E "grad_trace" || exit 1
for v in p2 p3; do
  echo "== trace kernel variant $v"
  GPX_LIB_PATH=$PWD/gaussian_process_amd/ab_libs/libgpx_$v.so timeout -k 10 200 python tools/trace_bench.py 2>&1 | grep -E "grad_trace" || exit 1
done
BLOCK=2048 bash tools/dist_trace.sh > $O/dtrace_b2048.log 2>&1 || { tail -20 $O/dtrace_b2048.log; exit 1; }
cat gpurun_out/dtrace_b2048/summary.txt
BLOCK=512 bash tools/dist_trace.sh > $O/dtrace_b512.log 2>&1 || { tail -20 $O/dtrace_b512.log; exit 1; }
cat gpurun_out/dtrace_b512/summary.txt
