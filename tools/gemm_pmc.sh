#!/bin/bash
# SQ counters of the dominant GEMM launch (one pass, 8 SQ counters + 1 GRBM)
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc_sq
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python3 $R/tools/gemm_launch.py 16384 2 > $R/gpurun_out/pmc_sq/log.txt 2>&1 || { tail -5 $R/gpurun_out/pmc_sq/log.txt; exit 1; }
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$R/gpurun_out/pmc_sq/sq_counter_collection.csv")) if "gemm" in r["Kernel_Name"]]
d={}
for r in rows: d.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in d.items(): print(k, v)
PY
