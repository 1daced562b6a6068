#!/bin/bash
# MFMA utilisation of the whole potrf at n = 65536 from PMC: sum of SQ_VALU_MFMA_BUSY_CYCLES over every
# kernel of the factorisation / (1024 SIMDs x busy GPU cycles).
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc_potrf
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_potrf -o pp --output-format csv -- \
  python3 $R/tools/potrf_trace.py 65536 > $R/gpurun_out/pmc_potrf/log.txt 2>&1 || { tail -5 $R/gpurun_out/pmc_potrf/log.txt; exit 1; }
python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open("$R/gpurun_out/pmc_potrf/pp_counter_collection.csv")))
byd=collections.OrderedDict()
for r in rows:
    d=byd.setdefault(int(r["Dispatch_Id"]),{"name":r["Kernel_Name"]})
    d[r["Counter_Name"]]=float(r["Counter_Value"])
ids=sorted(byd)
gram=max(i for i in ids if "gram_kernel" in byd[i]["name"])
sel=[byd[i] for i in ids if i>gram and ("gemm" in byd[i]["name"] or "leaf" in byd[i]["name"])]
mf=sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES",0) for d in sel); ga=sum(d.get("GRBM_GUI_ACTIVE",0) for d in sel)
print("potrf kernels %d: MFMA busy cycles %.4e, GPU active cycles (sum over 8 XCDs) %.4e" % (len(sel), mf, ga))
print("MFMA utilisation over the potrf kernels = %.4f" % (mf/(1024*ga/8)))
big=[d for d in sel if d.get("GRBM_GUI_ACTIVE",0)>8*2.0e7]
mfb=sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in big); gab=sum(d["GRBM_GUI_ACTIVE"] for d in big)
print("launches > ~8 ms: %d, utilisation %.4f, share of active cycles %.3f" % (len(big), mfb/(1024*gab/8), gab/ga))
with open("$R/gpurun_out/pmc_potrf/summary.txt","w") as f:
    f.write("potrf n=65536 PMC: kernels %d  SQ_VALU_MFMA_BUSY_CYCLES %.6e  GRBM_GUI_ACTIVE(sum XCD) %.6e  utilisation %.4f\n" % (len(sel), mf, ga, mf/(1024*ga/8)))
PY
