#!/bin/bash
# Full GPU suite under one limit; output goes to a file that grows as tests finish.
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -v --timeout 200 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -5 gpurun_out/gpu_tests.log
exit $rc
