#!/bin/bash
# kernel stats of the EP fit at n = 8192 (development): is the site loop bound by the kernels or by the gaps?
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/eptrace
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/eptrace -o ep --output-format csv -- \
  python3 $R/tools/ep_bench.py ${N:-8192} > $R/gpurun_out/eptrace/log.txt 2>&1 || { tail -5 $R/gpurun_out/eptrace/log.txt; exit 1; }
tail -2 $R/gpurun_out/eptrace/log.txt
head -8 $R/gpurun_out/eptrace/ep_kernel_stats.csv | cut -c1-160
find $R/gpurun_out/eptrace -name '*kernel_trace.csv' -size +20M -delete
