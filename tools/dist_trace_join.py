"""Join GPX_GEMM_LOG shapes with the kernel trace of the distributed step (tools/dist_trace.sh).  The engine
launches on two streams, so the join is by dispatch id (host launch order), not by start time."""
import csv, sys, collections
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
for r, s, ph in zip(gem, shapes, phase_of):
    if not ph.startswith("timed-"):
        continue
    M, N, K = int(s[1]), int(s[2]), int(s[3])
    lower = int(s[4].split("=")[1]); atri = int(s[5].split("=")[1]); btri = int(s[6].split("=")[1])
    fl = 2.0 * M * N * K * (0.5 if lower else 1.0) * (0.5 if (atri or btri) else 1.0)
    if len(s) > 9 and s[9] != "part=main":
        fl = 0.0          # tail / split-K parts of a launch logged with the launch's full shape: time only
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    kc = "K<=128" if K <= 128 else "K<=512" if K <= 512 else "K<=2048" if K <= 2048 else "K<=8192" if K <= 8192 else "K>8192"
    nc = "N<=128" if N <= 128 else "N<=512" if N <= 512 else "N<=2048" if N <= 2048 else "N>2048"
    key = "%-16s %-8s %-8s tri=%d%d low=%d" % (ph[6:], kc, nc, atri, btri, lower)
    c = cls.setdefault(key, [0, 0.0, 0.0])
    c[0] += 1; c[1] += t; c[2] += fl
    if ph == "timed-potrf" and K >= 2048 and fl > 0:
        big.setdefault((M, N, K), []).append(t)
tot = collections.defaultdict(lambda: [0.0, 0.0])
for k in sorted(cls):
    c = cls[k]
    print("%s launches %5d  time %8.2f ms  flops %8.3f T  rate %6.2f TF" % (k, c[0], c[1] * 1e3, c[2] / 1e12, c[2] / c[1] / 1e12))
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
print("potrf launches with K >= 2048 (main part), every 8th:")
for i, ((M, N, K), ts) in enumerate(big.items()):
    if i % 8 == 0:
        print("  %6d %5d %6d x%d  %9.1f us  %6.1f TF" % (M, N, K, len(ts), 1e6 * sum(ts) / len(ts), 2.0 * M * N * K / (sum(ts) / len(ts)) / 1e12))
