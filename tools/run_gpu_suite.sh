#!/bin/bash
# Full GPU suite, each step under its own limit; stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -15 gpurun_out/gpu_tests.log
exit $rc
