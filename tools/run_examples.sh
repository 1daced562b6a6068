#!/bin/bash
# run the four example scripts on the GPU box (development)
set -o pipefail
for f in regression_1d ard_regression_2d expectation_propagation distillation; do
  echo "== $f"; PYTHONPATH=$PWD timeout -k 10 120 python examples/$f.py 2>&1 | grep -v amdgpu.ids || exit 1
done
