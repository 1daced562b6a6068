"""Capacity check of the single-buffer estimator path: one fused LML + gradient evaluation
(gpx_gpr_lml_grad: Gram, potrf, solves, trtri, lauum into the strict upper triangle, traces) at an n whose
ONE n x n fp64 matrix fills a large part of the 288 GB of HBM (default n = 131072: 137 GB; with the
round-1 two-buffer layout this size did not fit).  Checked by size-independent properties: the noise
derivative against a central difference of two LML-only evaluations, and sampled entries of
(W^T W) Ky = I read back from the buffer (development tool; numbers quoted in DESIGN.md)."""
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
d = 8
rt = runtime(0)
rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
rng = np.random.default_rng(0)
Xh = rng.uniform(-3, 3, (n, d))
beta = 0.5 * np.logspace(-2, 0, d)
yh = np.sin(Xh @ np.linspace(0.3, 1.0, d)) + 0.3 * rng.standard_normal(n)
noise = 0.1
X, y = rt.upload_matrix(Xh), rt.upload_vector(yh)
K = rt.matrix(n, n)
dinv, vec, alpha, a = rt.dinv(n), rt.vector(5 * n), rt.vector(n), rt.vector(n)
_, bp = _keep = _lib.host_doubles(beta)
print("n=%d: one %.1f GB matrix; device memory in use %.1f GB" % (n, n * K.shape[1] * 8 / 1e9, torch.cuda.memory_allocated() / 1e9), flush=True)


def evaluate(nz, with_grad):
    lml = C.c_double()
    grad = (C.c_double * (2 + d))()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(rt.lib.gpx_gpr_lml_grad(rt.h, 1, P(X), n, X.shape[1], d, 1.0, bp, nz, P(y), P(K), K.shape[1], P(dinv),
                                       P(vec), P(alpha), P(a), C.byref(lml),
                                       C.cast(grad, C.POINTER(C.c_double)) if with_grad else None), "gpx_gpr_lml_grad")
    torch.cuda.synchronize()
    return lml.value, np.array(list(grad)), time.perf_counter() - t0


h = 1e-4
lp, _, tl = evaluate(noise + h, False)
lm, _, _ = evaluate(noise - h, False)
print("LML-only evaluation %.2f s (potrf n^3/3 at %.1f TFLOP/s incl. Gram and solves)" % (tl, n ** 3 / 3 / tl / 1e12), flush=True)
lml, g, tg = evaluate(noise, True)
print("fused LML + gradient %.2f s = %.1f TFLOP/s on n^3 flops; LML %.6f" % (tg, n ** 3 / tg / 1e12, lml), flush=True)
fd = (lp - lm) / (2 * h)
print("d LML / d noise: %.6f  central difference %.6f  (relative difference %.2e)" % (g[0], fd, abs(g[0] - fd) / abs(fd)), flush=True)
# sampled entries of Ky^-1 Ky = I with Ky^-1 read from the strict upper triangle (+ diagonal vector) of the buffer
kd = vec[4 * n: 5 * n]
worst = 0.0
for i in (0, 1, n // 3, n // 2 + 5, n - 2, n - 1):
    row = torch.empty(n, dtype=torch.float64, device=rt.dev)
    row[i] = kd[i]
    row[i + 1:] = K[i, i + 1: n]                 # Kinv[i][j], j > i
    row[:i] = K[:i, i]                           # Kinv[j][i], j < i (symmetric)
    for j in (0, i, n - 1):
        kcol = torch.from_numpy(np.exp(-0.5 * ((Xh - Xh[j]) ** 2 @ beta))).to(rt.dev)
        kcol[j] += noise
        worst = max(worst, abs(float(torch.dot(row, kcol).item()) - (1.0 if i == j else 0.0)))
print("max |(Ky^-1 Ky - I)_ij| over 18 sampled entries: %.2e" % worst, flush=True)
