#!/bin/bash
# Round-1 evidence: bench.py under rocprofv3 kernel trace, then two PMC passes on the dominant launch.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/bench -o r01 --output-format csv -- \
  python3 $R/bench.py --steps 2 --warmup 1 > $R/gpurun_out/prof/bench_r01.json.log 2>$R/gpurun_out/prof/bench_r01.err || { tail -5 $R/gpurun_out/prof/bench_r01.err; exit 1; }
tail -1 $R/gpurun_out/prof/bench_r01.json.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch -o r01 --output-format csv -- \
  python3 $R/tools/gemm_launch.py 16384 2 > $R/gpurun_out/prof/pmc_fetch.log 2>&1 || { tail -5 $R/gpurun_out/prof/pmc_fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write -o r01 --output-format csv -- \
  python3 $R/tools/gemm_launch.py 16384 2 > $R/gpurun_out/prof/pmc_write.log 2>&1 || { tail -5 $R/gpurun_out/prof/pmc_write.log; exit 1; }
# keep only the small summaries (the full kernel trace is large)
find $R/gpurun_out/prof -name '*kernel_trace.csv' -size +8M -delete
ls -la $R/gpurun_out/prof/*
