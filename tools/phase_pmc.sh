#!/bin/bash
# MFMA utilisation launch by launch over potrf + trtri + lauum at n (default 65536): which of the long launches
# fall short, and by how much (development tool).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_phase
rm -rf $O; mkdir -p $O
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O -o pp --output-format csv -- \
  python3 $R/tools/potrf_trace.py ${N:-65536} all > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - <<PY
import csv, collections, glob, re
f = glob.glob("$O/**/pp_counter_collection.csv", recursive=True)[0]
byd = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = byd.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r["Grid_Size"]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
rows = []
for i, d in byd.items():
    if "gemm_f64_kernel" not in d["name"] or d.get("GRBM_GUI_ACTIVE", 0) < 8 * 2.0e7:
        continue
    m = re.search(r"gemm_f64_kernel<(\d), (\d), (\d), (\d)>", d["name"])
    util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8)
    rows.append((i, "A%s B%s" % (("KC", "RC")[int(m.group(1))], ("KC", "RC")[int(m.group(2))]), int(d["grid"]) // 256, d["GRBM_GUI_ACTIVE"] / 8 / 2.4e6, util))
print("dispatch  layouts  workgroups   ~ms   MFMA busy")
for r in rows:
    print("%7d  %s  %9d  %6.1f  %.4f" % r)
open("$O/summary.txt", "w").write("\n".join("%d %s %d %.1f %.4f" % r for r in rows) + "\n")
PY
