#!/bin/bash
# kernel trace of the single-right-hand-side solves at n (default 65536): time by kernel of the LAST forward and the
# LAST backward solve of tools/trsv_bench.py (block-inverse path from n = 8192 on; GPX_TRSV_BLOCKINV_MIN=0: recursion)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/trsv_trace
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $O -o tt --output-format csv -- python3 $R/tools/trsv_bench.py ${N:-65536} > $O/run.log 2>&1
grep -E "^trsv" $O/run.log
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$O/**/tt_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def kname(r):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:44]
mine = [r for r in rows if re.search(r"trsv|trmv|gemvt_strip|binv_init|vec_copy|gemm_f64", r["Kernel_Name"])]
# a solve = run of such kernels starting with vec_copy (gather) ... ending with vec_copy (scatter); the prepare precedes it
solves, cur = [], []
for r in mine:
    cur.append(r)
    if "vec_copy" in r["Kernel_Name"] and len([x for x in cur if "vec_copy" in x["Kernel_Name"]]) >= 2 and \
       ("step" in cur[-2]["Kernel_Name"] or "strip" in cur[-2]["Kernel_Name"] or "trmv" in cur[-2]["Kernel_Name"] or "gemvn" in cur[-2]["Kernel_Name"]):
        solves.append(cur); cur = []
print("solves found:", len(solves))
for sv in solves[-4:-3] + solves[-1:]:
    t0 = min(int(r["Start_Timestamp"]) for r in sv); t1 = max(int(r["End_Timestamp"]) for r in sv)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sv)
    print("solve: %d kernels, span %.3f ms, busy %.3f ms, gaps %.3f ms" % (len(sv), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    by = collections.OrderedDict()
    for r in sv:
        d = by.setdefault(kname(r), [0, 0]); d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, d in by.items():
        print("   %-46s %5d launches %8.3f ms  avg %7.2f us" % (k, d[0], d[1] / 1e6, d[1] / d[0] / 1e3))
PY
rm -f $O/*/tt_kernel_trace.csv $O/tt_kernel_trace.csv
