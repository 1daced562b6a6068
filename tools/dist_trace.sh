#!/bin/bash
# Launch classes of the 2-D block-cyclic step on ONE rank over RCCL (development tool): GPX_GEMM_LOG shapes of
# every GEMM launch joined with a rocprofv3 kernel trace by dispatch id.  BLOCK=<tile> (default 2048).
# Output: gpurun_out/dtrace_b$BLOCK/summary.txt
export TMPDIR=/tmp GPX_GEMM_LOG=1 GPX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
R=$PWD
B=${BLOCK:-2048}
D=$R/gpurun_out/dtrace_b$B
mkdir -p $D
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace -d $D -o dt --output-format csv -- python3 $R/bench.py --gpus 1 \
  --steps 1 --warmup 1 --block $B --no-cpu-baseline --no-public-api > $D/stdout.log 2> $D/stderr.log || { tail -5 $D/stderr.log; exit 1; }
grep -E 'GEMMLOG|PHASE' $D/stderr.log > $D/gemmlog.txt
python3 $R/tools/dist_trace_join.py $D > $D/summary.txt || { tail -5 $D/summary.txt; exit 1; }
rm -f $D/dt_kernel_trace.csv $D/stderr.log
cat $D/summary.txt
