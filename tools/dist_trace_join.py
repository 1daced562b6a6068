"""Join GPX_GEMM_LOG shapes with the kernel trace of the distributed step (tools/dist_trace.sh).  The engine
launches on two streams, so the join is by dispatch id (host launch order), not by start time."""
import csv, sys, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_common import parse_shape, merge_parts, PEAK_TF
d = sys.argv[1]
shapes, phase_of, phase, nstart = [], [], "warmup", 0
for l in open(d + "/gemmlog.txt"):
    if l.startswith("PHASE"):
        name = l.split()[1]
        if name == "after-start":
            nstart += 1
        order = ["after-start", "after-gram", "after-potrf", "after-solve_lml", "after-predict", "after-grad"]
        nxt = ["gram", "potrf", "solve_lml", "predict", "grad", "between"]
        phase = ("timed-" if nstart >= 2 else "warm-") + nxt[order.index(name)]
    elif l.startswith("GEMMLOG"):
        shapes.append(l.split()); phase_of.append(phase)
rows = list(csv.DictReader(open(d + "/dt_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
gem = [r for r in rows if "gemm_f64_kernel" in r["Kernel_Name"]]
print("gemm launches in trace %d, in log %d" % (len(gem), len(shapes)))
cls = collections.OrderedDict()
big = collections.OrderedDict()
assert len(gem) == len(shapes), "trace and log disagree: the join would be wrong"
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
# main / tail / split-K parts of one launch merged: time added, flops once (trace_common.py)
logical = merge_parts([(parse_shape(s), dur(r), ph) for r, s, ph in zip(gem, shapes, phase_of)])
for sh, t, fl, nk, ph in logical:
    if not ph.startswith("timed-"):
        continue
    M, N, K = sh["M"], sh["N"], sh["K"]
    kc = "K<=128" if K <= 128 else "K<=512" if K <= 512 else "K<=2048" if K <= 2048 else "K<=8192" if K <= 8192 else "K>8192"
    nc = "N<=128" if N <= 128 else "N<=512" if N <= 512 else "N<=2048" if N <= 2048 else "N>2048"
    key = "%-16s %-8s %-8s tri=%d%d low=%d%s" % (ph[6:], kc, nc, sh["atri"], sh["btri"], sh["lower"], " grouped" if sh["cb"] else "")
    c = cls.setdefault(key, [0, 0.0, 0.0])
    c[0] += 1; c[1] += t; c[2] += fl
    if ph == "timed-potrf" and K >= 2048:
        big.setdefault((M, N, K, fl), []).append(t)
tot = collections.defaultdict(lambda: [0.0, 0.0])
for k in sorted(cls):
    c = cls[k]
    print("%s launches %5d  time %8.2f ms  flops %8.3f T  rate %6.2f TF" % (k, c[0], c[1] * 1e3, c[2] / 1e12, c[2] / c[1] / 1e12))
    assert c[2] / c[1] / 1e12 <= PEAK_TF, "a class above the fp64 MFMA peak: flop model or join wrong"
    ph = k.split()[0]
    tot[ph][0] += c[1]; tot[ph][1] += c[2]
for ph, (t, f) in tot.items():
    print("phase %-10s gemm time %8.2f ms  flops %8.3f T  avg %6.2f TF" % (ph, t * 1e3, f / 1e12, f / t / 1e12 if t else 0))
other = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if "gemm_f64_kernel" not in r["Kernel_Name"]:
        nme = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
        other[nme][0] += 1
        other[nme][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
print("other kernels (warm-up + timed step):")
for k, v in sorted(other.items(), key=lambda kv: -kv[1][1])[:12]:
    print("  %-52s %6d %9.2f ms" % (k, v[0], v[1]))
lk = [r for r in rows if "leaf_blocked_kernel" in r["Kernel_Name"] or "potrf_chain" in r["Kernel_Name"]]
if lk:
    ts = sorted(dur(r) for r in lk)
    print("leaf / chain kernels: %d, average %.1f us, median %.1f us, max %.1f us" % (len(ts), 1e6 * sum(ts) / len(ts), 1e6 * ts[len(ts) // 2], 1e6 * ts[-1]))
print("potrf launches with K >= 2048 (parts merged), every 8th:")
for i, ((M, N, K, fl), ts) in enumerate(big.items()):
    if i % 8 == 0:
        print("  %6d %5d %6d x%d  %9.1f us  %6.1f TF" % (M, N, K, len(ts), 1e6 * sum(ts) / len(ts), fl / (sum(ts) / len(ts)) / 1e12))
