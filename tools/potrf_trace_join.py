"""Join GPX_GEMM_LOG shapes with the kernel trace (same order on one stream) and summarise GEMM
efficiency by size class."""
import csv, sys, collections
d = sys.argv[1]
shapes, phase_of, phase = [], [], "potrf"
for l in open(d + "/gemmlog.txt"):
    if l.startswith("PHASE"):
        phase = l.split()[1]
    elif l.startswith("GEMMLOG"):
        shapes.append(l.split()); phase_of.append(phase)
rows = list(csv.DictReader(open(d + "/pt_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# potrf's kernels: after the (single) gram kernel
gi = max(i for i, nme in enumerate(names) if "gram_kernel" in nme)
gem = [r for r in rows[gi + 1:] if "gemm_f64_kernel" in r["Kernel_Name"]]
leaf = [r for r in rows[gi + 1:] if "leaf" in r["Kernel_Name"]]
print("gemm launches in trace %d, in log %d; leaves %d" % (len(gem), len(shapes), len(leaf)))
cls = collections.OrderedDict()
tot_t = tot_f = 0.0
for r, s, ph in zip(gem, shapes, phase_of):
    M, N, K = int(s[1]), int(s[2]), int(s[3])
    lower = int(s[4].split("=")[1]); atri = int(s[5].split("=")[1]); btri = int(s[6].split("=")[1])
    fl = 2.0 * M * N * K * (0.5 if lower else 1.0) * (0.5 if (atri or btri) else 1.0)
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    key = "K<=128" if K <= 128 else "K<=512" if K <= 512 else "K<=2048" if K <= 2048 else "K<=8192" if K <= 8192 else "K>8192"
    key = ph + " " + key + " " + s[8] + (" " + s[9] if len(s) > 9 and not s[9].endswith("main") else "")
    c = cls.setdefault(key, [0, 0.0, 0.0])
    c[0] += 1; c[1] += t; c[2] += fl
    tot_t += t; tot_f += fl
for k in sorted(cls):
    c = cls[k]
    print("%-16s launches %5d  time %8.2f ms  flops %8.3f T  rate %6.2f TF" % (k, c[0], c[1] * 1e3, c[2] / 1e12, c[2] / c[1] / 1e12))
lt = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in leaf) * 1e-9
span = (int(rows[-1]["End_Timestamp"]) - int(rows[gi + 1]["Start_Timestamp"])) * 1e-9
print("gemm total %.1f ms (%.2f TF avg), leaves %.1f ms, span %.1f ms -> gaps/other %.1f ms" % (tot_t * 1e3, tot_f / tot_t / 1e12, lt * 1e3, span * 1e3, (span - tot_t - lt) * 1e3))
# the long launches one by one
print("launches with K >= 1024 in potrf:")
for r, s, ph in zip(gem, shapes, phase_of):
    M, N, K = int(s[1]), int(s[2]), int(s[3])
    if ph == "potrf" and K >= 1024:
        lower = int(s[4].split("=")[1])
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        print("  %5d %5d %5d %s  %9.1f us  %6.1f TF" % (M, N, K, " ".join(s[4:]), t, 2.0 * M * N * K * (0.5 if lower else 1.0) / t / 1e6))
