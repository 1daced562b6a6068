#!/bin/bash
set -o pipefail
O=gpurun_out/s6
mkdir -p $O
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread -m gpu tests/test_gpu_primitives.py -k "block_inverses or potrf_trsm" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for n in 8192 16384 65536; do
echo "== trsv n=$n block inverses";  timeout -k 10 200 python tools/trsv_bench.py $n 2>&1 | grep -E "^trsv" || exit 1
done
