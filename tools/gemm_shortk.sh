#!/bin/bash
export TMPDIR=/tmp GPX_GEMM_LOG=1
R=$PWD
mkdir -p $R/gpurun_out/shortk
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/shortk -o sk --output-format csv -- python3 $R/tools/gemm_shortk.py 2> $R/gpurun_out/shortk/stderr.log || { tail -5 $R/gpurun_out/shortk/stderr.log; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/shortk/sk_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
gem = [r for r in rows if "gemm_f64_kernel" in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]]
logs = [l.split() for l in open("$R/gpurun_out/shortk/stderr.log") if l.startswith("GEMMLOG")]
print(len(gem), len(logs))
import collections
acc = collections.OrderedDict()
for r, s in zip([g for g in gem if "gemm_f64" in g["Kernel_Name"]], logs):
    key = (s[1], s[2], s[3], s[8], s[9])
    acc.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    print("%5s %5s %5s %s %-10s  min %7.2f us  (%s)" % (k[0], k[1], k[2], k[3], k[4], min(v), " ".join("%.1f" % x for x in v)))
PY
