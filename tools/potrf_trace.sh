#!/bin/bash
export TMPDIR=/tmp GPX_GEMM_LOG=1
R=$PWD
mkdir -p $R/gpurun_out/ptrace
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/ptrace -o pt --output-format csv -- python3 $R/tools/potrf_trace.py ${N:-65536} all 2> $R/gpurun_out/ptrace/stderr.log
grep -A100000 POTRF-BEGIN $R/gpurun_out/ptrace/stderr.log | grep -E 'GEMMLOG|PHASE' > $R/gpurun_out/ptrace/gemmlog.txt
python3 - <<'PY'
import csv, os
R=os.environ.get("R_DIR")
PY
cut -d, -f1-20 $R/gpurun_out/ptrace/pt_kernel_trace.csv | head -2
python3 $R/tools/potrf_trace_join.py $R/gpurun_out/ptrace
