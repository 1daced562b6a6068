#!/bin/bash
set -o pipefail
O=gpurun_out/s7
mkdir -p $O
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/gpu_tests.log 2>&1
rc=$?
tail -5 $O/gpu_tests.log
exit $rc
