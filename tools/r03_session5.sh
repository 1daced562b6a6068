#!/bin/bash
set -o pipefail
O=gpurun_out/s5
mkdir -p $O
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_gpu_primitives.py -k "block_inverses or potrf_trsm or dtheta or non_finite" tests/test_gpu_regression.py tests/test_gpu_ep.py > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for n in 8192 16384 65536; do
echo "== trsv n=$n recursion (GPX_TRSV_BLOCKINV_MIN=0)"; GPX_TRSV_BLOCKINV_MIN=0 timeout -k 10 200 python tools/trsv_bench.py $n 2>&1 | grep -E "^trsv" || exit 1
echo "== trsv n=$n block inverses";  timeout -k 10 200 python tools/trsv_bench.py $n 2>&1 | grep -E "^trsv" || exit 1
done
timeout -k 10 100 python tools/nonpd_timing.py 2>&1 | grep -v amdgpu | tail -5
