#!/bin/bash
# kernel trace of the single-right-hand-side solves at n = 65536: time by kernel and stream, and the span of each solve
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/trsv_trace
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $O -o tt --output-format csv -- python3 $R/tools/trsv_bench.py ${N:-65536} > $O/run.log 2>&1
tail -2 $O/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$O/**/tt_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "trsv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into solves: a gap > 1 ms between consecutive trsv kernels starts a new solve
import re
def kname(r):
    m = re.search(r"(trsv_\w+)", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:40]
nleaf = (int("${N:-65536}") + 255) // 256
solves, cur, leaves = [], [], 0
for r in rows:
    if "leaf" in kname(r):
        if leaves == nleaf:
            solves.append(cur); cur = []; leaves = 0
        leaves += 1
    cur.append(r)
solves.append(cur)
for si, sv in enumerate([solves[2], solves[5]] if len(solves) >= 6 else solves):
    t0 = min(int(r["Start_Timestamp"]) for r in sv); t1 = max(int(r["End_Timestamp"]) for r in sv)
    print("solve %d: %d kernels, span %.3f ms" % (si, len(sv), (t1 - t0) / 1e6))
    by = collections.OrderedDict()
    for r in sv:
        k = (kname(r), r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))
        d = by.setdefault(k, [0, 0])
        d[0] += 1; d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, d in by.items():
        print("   %-28s stream %s queue %s: %5d launches %8.3f ms  avg %7.2f us" % (k[0], k[1], k[2], d[0], d[1] / 1e6, d[1] / d[0] / 1e3))
    lf = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sv if "leaf" in kname(r)]
    q = max(1, len(lf) // 8)
    print("   leaf us by eighth of the solve: " + " ".join("%.1f" % (sum(lf[i:i + q]) / len(lf[i:i + q]) / 1e3) for i in range(0, len(lf), q)))
    # chain gaps: time between consecutive kernels on the first kernel's stream
    sid = sv[0].get("Stream_Id")
    ch = [r for r in sv if r.get("Stream_Id") == sid]
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(ch[:-1], ch[1:])]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ch)
    print("   chain stream %s: %d kernels busy %.3f ms, gaps %.3f ms (max %.1f us, %d gaps > 20 us)" % (sid, len(ch), busy / 1e6, sum(gaps) / 1e6, max(gaps) / 1e3, sum(g > 20000 for g in gaps)))
PY
rm -f $O/*/tt_kernel_trace.csv $O/tt_kernel_trace.csv
