#!/bin/bash
# HBM bytes of the Gram kernel from PMC (WRITE_SIZE in one pass, FETCH_SIZE in another) next to its duration.
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc_gram
cd /tmp
for c in WRITE_SIZE FETCH_SIZE; do
timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_gram/$c -o g --output-format csv -- \
  python3 $R/tools/gram_bench.py 65536 > $R/gpurun_out/pmc_gram/$c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_gram/$c.log; exit 1; }
done
python3 - <<PY
import csv
out=[]
for c in ("WRITE_SIZE","FETCH_SIZE"):
    rows=[r for r in csv.DictReader(open("$R/gpurun_out/pmc_gram/%s/g_counter_collection.csv"%c)) if "gram_kernel" in r["Kernel_Name"]]
    tr={r["Dispatch_Id"]:r for r in csv.DictReader(open("$R/gpurun_out/pmc_gram/%s/g_kernel_trace.csv"%c))}
    for r in rows:
        t=tr.get(r["Dispatch_Id"]); ms=(int(t["End_Timestamp"])-int(t["Start_Timestamp"]))/1e6 if t else float("nan")
        out.append("%s dispatch %s: %.1f KiB = %.3f GB, kernel %.3f ms" % (c, r["Dispatch_Id"], float(r["Counter_Value"]), float(r["Counter_Value"])*1024/1e9, ms))
print("\n".join(out))
open("$R/gpurun_out/pmc_gram/summary.txt","w").write("\n".join(out)+"\n")
PY
