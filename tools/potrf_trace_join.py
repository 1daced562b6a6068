"""Join GPX_GEMM_LOG shapes with the kernel trace (same order on one stream) and summarise GEMM
efficiency by size class.  Flop model and the merging of main / tail / split-K parts: trace_common.py."""
import csv, sys, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_common import parse_shape, merge_parts, PEAK_TF
d = sys.argv[1]
shapes, phase_of, phase = [], [], "potrf"
for l in open(d + "/gemmlog.txt"):
    if l.startswith("PHASE"):
        phase = l.split()[1]
    elif l.startswith("GEMMLOG"):
        shapes.append(parse_shape(l.split())); phase_of.append(phase)
rows = list(csv.DictReader(open(d + "/pt_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# potrf's kernels: after the (single) gram kernel
gi = max(i for i, nme in enumerate(names) if "gram_kernel" in nme)
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
gem = [r for r in rows[gi + 1:] if "gemm_f64_kernel" in r["Kernel_Name"]]
leaf = [r for r in rows[gi + 1:] if "leaf" in r["Kernel_Name"] or "potrf_chain" in r["Kernel_Name"]]
red = [r for r in rows[gi + 1:] if "splitk_reduce" in r["Kernel_Name"]]
print("gemm kernels in trace %d, in log %d; leaf / chain kernels %d; split-K reduce kernels %d" % (len(gem), len(shapes), len(leaf), len(red)))
assert len(gem) == len(shapes), "trace and log disagree: the join would be wrong"
logical = merge_parts([(s, dur(r), ph) for r, s, ph in zip(gem, shapes, phase_of)])
cls = collections.OrderedDict()
tot_t = tot_f = 0.0
for sh, t, fl, nk, ph in logical:
    K = sh["K"]
    key = "K<=128" if K <= 128 else "K<=512" if K <= 512 else "K<=2048" if K <= 2048 else "K<=8192" if K <= 8192 else "K>8192"
    key = ph + " " + key + " cfg=" + sh["cfg"]
    c = cls.setdefault(key, [0, 0.0, 0.0, 0])
    c[0] += 1; c[1] += t; c[2] += fl; c[3] += nk
    tot_t += t; tot_f += fl
worst = 0.0
for k in sorted(cls):
    c = cls[k]
    rate = c[2] / c[1] / 1e12
    worst = max(worst, rate)
    print("%-22s launches %5d (kernels %5d)  time %8.2f ms  flops %8.3f T  rate %6.2f TF" % (k, c[0], c[3], c[1] * 1e3, c[2] / 1e12, rate))
assert worst <= PEAK_TF, "a class above the fp64 MFMA peak: flop model or join wrong"
lt = sum(dur(r) for r in leaf)
rt_ = sum(dur(r) for r in red)
span = (int(rows[-1]["End_Timestamp"]) - int(rows[gi + 1]["Start_Timestamp"])) * 1e-9
print("gemm total %.1f ms (%.2f TF avg over %.3f Tflop), leaves %.1f ms, split-K reduces %.1f ms, span %.1f ms -> gaps/other %.1f ms"
      % (tot_t * 1e3, tot_f / tot_t / 1e12, tot_f / 1e12, lt * 1e3, rt_ * 1e3, span * 1e3, (span - tot_t - lt - rt_) * 1e3))
byph = collections.OrderedDict()
for sh, t, fl, nk, ph in logical:
    b = byph.setdefault(ph, [0.0, 0.0]); b[0] += t; b[1] += fl
for ph, (t, f) in byph.items():
    print("phase %-6s gemm time %8.2f ms  flops %8.3f T  avg %6.2f TF" % (ph, t * 1e3, f / 1e12, f / t / 1e12))
# the long launches one by one
print("launches with K >= 1024 in potrf (parts merged):")
for sh, t, fl, nk, ph in logical:
    if ph == "potrf" and sh["K"] >= 1024:
        print("  %5d %5d %5d lower=%d cfg=%s kernels=%d  %9.1f us  %6.1f TF" % (sh["M"], sh["N"], sh["K"], sh["lower"], sh["cfg"], nk, t * 1e6, fl / t / 1e12))
