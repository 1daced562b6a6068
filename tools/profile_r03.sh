#!/bin/bash
# Round-3 evidence: bench.py under rocprofv3 kernel trace + stats, then separate PMC passes (FETCH_SIZE /
# WRITE_SIZE) over the dominant GEMM launch and the Gram kernel, and the MFMA-busy pass over the potrf.
# Writes the small summaries to gpurun_out/prof3/ (copied into profiles/r03_* afterwards).
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof3
mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/bench -o r03 --output-format csv -- \
  python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-public-api > $O/bench_r03.json.log 2>$O/bench_r03.err || { tail -5 $O/bench_r03.err; exit 1; }
tail -1 $O/bench_r03.json.log | cut -c1-400
python3 - <<PY
import csv, glob
f = glob.glob("$O/bench/**/*kernel_stats.csv", recursive=True)
print("kernel stats:", f)
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d $O/pmc_$c -o g --output-format csv -- \
    python3 $R/tools/gemm_launch.py 16384 2 > $O/pmc_$c.log 2>&1 || { tail -5 $O/pmc_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, json
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm_f64_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    val[c] = [float(r["Counter_Value"]) for r in rows]
fetch = sum(val["FETCH_SIZE"]) / len(val["FETCH_SIZE"]); write = sum(val["WRITE_SIZE"]) / len(val["WRITE_SIZE"])
out = {"kernel": "gemm_f64_kernel<KC,KC,4,4>", "shape": "M=N=K=16384, D = C - A B^T",
       "fetch_size_kib": fetch, "write_size_kib": write, "per_launch_fetch_kib": val["FETCH_SIZE"], "per_launch_write_kib": val["WRITE_SIZE"],
       "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
       "formula": "(2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md)",
       "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/gemm_launch.py 16384 2"}
json.dump(out, open("$O/r03_gemm_launch_pmc.json", "w"), indent=1)
print(out)
PY
cd $R && bash tools/gram_pmc.sh > $O/gram_pmc.log 2>&1; tail -4 $O/gram_pmc.log
cp $R/gpurun_out/pmc_gram/summary.txt $O/r03_gram_pmc_bytes.txt 2>/dev/null
cd $R && bash tools/potrf_pmc.sh > $O/potrf_pmc.log 2>&1; tail -3 $O/potrf_pmc.log
cp $R/gpurun_out/pmc_potrf/summary.txt $O/r03_potrf_pmc_mfma_util.txt 2>/dev/null
find $O -name '*kernel_trace.csv' -size +8M -delete
find $O -name '*counter_collection.csv' -size +8M -delete
ls $O
