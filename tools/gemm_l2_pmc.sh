#!/bin/bash
# L2 behaviour of the dominant GEMM launch from PMC, two passes of their own: hit / miss counts of the XCD L2s,
# and the read requests the L2s send to the fabric by size (TCC_EA_RDREQ counts 64-byte requests unless flagged 32B).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_l2
rm -rf $O; mkdir -p $O
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/hit -o l2 --output-format csv -- python3 $R/tools/gemm_launch.py ${1:-16384} 2 > $O/hit.log 2>&1 || { tail -5 $O/hit.log; exit 1; }
true
python3 - <<PY
import csv, glob
out = []
for sub in ("hit",):
    f = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)[0]
    d = {}
    for r in csv.DictReader(open(f)):
        if "gemm_f64_kernel" in r["Kernel_Name"]:
            d.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in sorted(d.items(), key=lambda kv: int(kv[0])):
        line = "%s dispatch %s: " % (sub, k) + "  ".join("%s %.4e" % kv for kv in sorted(v.items()))
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            line += "  -> L2 hit rate %.4f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]))
        if "TCC_EA_RDREQ_sum" in v:
            b32 = v.get("TCC_EA_RDREQ_32B_sum", 0.0)
            line += "  -> fabric read bytes %.2f GB (64 B x (RDREQ - 32B) + 32 B x 32B)" % ((64 * (v["TCC_EA_RDREQ_sum"] - b32) + 32 * b32) / 1e9)
        out.append(line)
print("\n".join(out))
open("$O/summary.txt", "w").write("\n".join(out) + "\n")
PY
