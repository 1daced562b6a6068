#!/bin/bash
# SQ counters of grad_trace_kernel (and gram_kernel) at n = 65536: how busy the VALU is (development tool)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_trace
rm -rf $O; mkdir -p $O
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE \
  -d $O -o tp --output-format csv -- python3 $R/tools/trace_bench.py 65536 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/tp_counter_collection.csv", recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "grad_trace_kernel" in r["Kernel_Name"] or "gram_kernel" in r["Kernel_Name"]:
        d.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:60]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
seen = {}
for (did, name), v in d.items():
    key = name
    seen[key] = (did, v)      # keep the last dispatch of each kernel
for name, (did, v) in seen.items():
    ga = v.get("GRBM_GUI_ACTIVE", 0) / 8        # per-XCD sum -> cycles
    print(name, "dispatch", did)
    print("   GPU active cycles %.3e (%.3f ms at 2.4 GHz)" % (ga, ga / 2.4e6))
    for k in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS"):
        print("   %-22s %.4e" % (k, v.get(k, float("nan"))))
    if ga:
        print("   VALU instructions per SIMD-cycle: %.3f  (1024 SIMDs); active-VALU cycles / (1024 x GPU cycles) = %.3f"
              % (v.get("SQ_INSTS_VALU", 0) / (1024 * ga), v.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * ga)))
PY
