"""Times gpx_grad_trace / gpx_gram / the two single-RHS solves at full size.  python tools/trace_bench.py [n]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.datasets import synthetic_ard
from gaussian_process_amd.device import ptr, runtime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rt = runtime(0)
rt.bind_stream()
X, y, _, beta = synthetic_ard(n, m=4, d=8, seed=0)
dX, dy = rt.upload_matrix(X), rt.upload_vector(y)
A = rt.matrix(n, n)
dinv = rt.dinv(n)
_, bp = keep = _lib.host_doubles(beta)
ld = A.shape[1]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)


lower_bytes = 4.0 * n * (n + 1)
t = timed(lambda: rt.lib.gpx_gram(rt.h, 1, ptr(dX), n, dX.shape[1], None, 0, 0, 8, 1.0, bp, 0.1, ptr(A), ld, 1))
print("gram lower   min %.3f ms avg %.3f ms  %.2f TB/s" % (t[0], t[1], lower_bytes / t[0] / 1e9), flush=True)
out = (C.c_double * 10)()
t = timed(lambda: rt.lib.gpx_grad_trace(rt.h, 1, ptr(dX), n, dX.shape[1], 8, 1.0, bp, ptr(dy), ptr(A), ld, None,
                                        C.cast(out, C.POINTER(C.c_double))))
print("grad_trace   min %.3f ms avg %.3f ms  %.2f TB/s (incl. final reduction + fetch)" % (t[0], t[1], lower_bytes / t[0] / 1e9), flush=True)
info = rt.lib.gpx_potrf(rt.h, n, ptr(A), ld, ptr(dinv))
assert info == 0, info
v = torch.zeros((n, 2), dtype=torch.float64, device=rt.dev)
v[:, 0].copy_(dy[:n])
for trans in (0, 1):
    t = timed(lambda: rt.lib.gpx_trsm(rt.h, trans, n, 1, ptr(A), ld, ptr(dinv), ptr(v), 2))
    print("trsv trans=%d min %.3f ms avg %.3f ms  %.2f TB/s" % (trans, t[0], t[1], lower_bytes / t[0] / 1e9), flush=True)
