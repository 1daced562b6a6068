"""Capacity check: lower Gram + Cholesky of ONE n x n fp64 matrix that fills a large part of the 288 GB of
HBM (default n = 131072: 137 GB), verified on sampled entries of L L^T against K recomputed from X
(development tool; the number is quoted in DESIGN.md)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rt = runtime(0)
rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
rng = np.random.default_rng(0)
Xh = rng.uniform(-3, 3, (n, 8))
beta = 0.5 * np.logspace(-2, 0, 8)
noise = 0.1
X = rt.upload_matrix(Xh)
K = rt.matrix(n, n)
dinv = rt.dinv(n)
_, bp = _keep = _lib.host_doubles(beta)
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.check(rt.lib.gpx_gram(rt.h, 1, P(X), n, X.shape[1], None, 0, 0, 8, 1.0, bp, noise, P(K), K.shape[1], 1), "gram")
torch.cuda.synchronize()
t1 = time.perf_counter()
_lib.check(rt.lib.gpx_potrf(rt.h, n, P(K), K.shape[1], P(dinv)), "potrf")
torch.cuda.synchronize()
t2 = time.perf_counter()
ld = C.c_double()
_lib.check(rt.lib.gpx_logdet(rt.h, n, P(K), K.shape[1], C.byref(ld)), "logdet")
worst = 0.0
pairs = [(0, 0), (n - 1, n - 1), (n - 1, 0), (n // 2 + 77, n // 2 - 5), (n - 3, n - 130), (n // 3, n // 7),
         (n - 1, n // 2), (n // 2, n // 2)]
for i, j in pairs:
    got = float(torch.dot(K[i, : j + 1], K[j, : j + 1]).item())
    ref = float(np.exp(-0.5 * np.sum(beta * (Xh[i] - Xh[j]) ** 2))) + (noise if i == j else 0.0)
    worst = max(worst, abs(got - ref))
print("n=%d (%.1f GB): gram %.1f ms (%.2f TB/s), potrf %.2f s = %.1f TFLOP/s, logdet %.6f, max |(L L^T - Ky)_ij| over %d sampled entries %.2e"
      % (n, n * K.shape[1] * 8 / 1e9, 1e3 * (t1 - t0), 4.0 * n * (n + 1) / (t1 - t0) / 1e12, t2 - t1,
         n ** 3 / 3 / (t2 - t1) / 1e12, ld.value, len(pairs), worst), flush=True)

# round 3: the two single-right-hand-side solves at this size (block-inverse path, 1 GB of workspace at n = 131072),
# checked on sampled rows of Ky x = b with the rows of Ky recomputed from X on the host
b = rng.standard_normal(n)
v = rt.upload_vector(b)
torch.cuda.synchronize()
t3 = time.perf_counter()
_lib.check(rt.lib.gpx_trsm(rt.h, 0, n, 1, P(K), K.shape[1], P(dinv), P(v), 1), "trsm fwd")
_lib.check(rt.lib.gpx_trsm(rt.h, 1, n, 1, P(K), K.shape[1], P(dinv), P(v), 1), "trsm bwd")
torch.cuda.synchronize()
t4 = time.perf_counter()
x = rt.download(v[:n])
worst_r = 0.0
for i in (0, 1, n // 3, n // 2 + 77, n - 2, n - 1):
    row = np.exp(-0.5 * ((Xh - Xh[i]) ** 2 @ beta))
    row[i] += noise
    worst_r = max(worst_r, abs(float(row @ x) - b[i]))
print("two single-RHS solves %.1f ms (first call: includes the workspace allocation and the block inverses), "
      "max |(Ky x - b)_i| over 6 sampled rows %.2e (|x|_inf %.2e)" % (1e3 * (t4 - t3), worst_r, np.abs(x).max()), flush=True)
