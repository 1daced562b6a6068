#!/bin/bash
# Launch classes of the 2-D block-cyclic step on ONE rank over RCCL (development tool): GPX_GEMM_LOG shapes of
# every GEMM launch joined with a rocprofv3 kernel trace by dispatch order.  Output: gpurun_out/dtrace/summary.txt
export TMPDIR=/tmp GPX_GEMM_LOG=1 GPX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
R=$PWD
mkdir -p $R/gpurun_out/dtrace
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace -d $R/gpurun_out/dtrace -o dt --output-format csv -- python3 $R/bench.py --gpus 1 \
  --steps 1 --warmup 1 --no-cpu-baseline --no-public-api > $R/gpurun_out/dtrace/stdout.log 2> $R/gpurun_out/dtrace/stderr.log || { tail -5 $R/gpurun_out/dtrace/stderr.log; exit 1; }
grep -E 'GEMMLOG|PHASE' $R/gpurun_out/dtrace/stderr.log > $R/gpurun_out/dtrace/gemmlog.txt
python3 $R/tools/dist_trace_join.py $R/gpurun_out/dtrace > $R/gpurun_out/dtrace/summary.txt
gzip -f $R/gpurun_out/dtrace/dt_kernel_trace.csv
cat $R/gpurun_out/dtrace/summary.txt
