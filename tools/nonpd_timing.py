"""Times gpx_potrf on a PD and on an indefinite matrix of the same size, and the fused evaluation at a
non-PD theta.  python tools/nonpd_timing.py [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import ARD, GaussianProcessRegression, _lib
from gaussian_process_amd.datasets import synthetic_ard
from gaussian_process_amd.device import ptr, runtime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rt = runtime(0)
rt.bind_stream()
X, y, _, beta = synthetic_ard(n, m=4, d=8, seed=4)
dX = rt.upload_matrix(X)
A = rt.matrix(n, n)
dinv = rt.dinv(n)
_, bp = keep = _lib.host_doubles(beta)
for noise in (0.1, -0.5, 0.1, -0.5):
    rt.lib.gpx_gram(rt.h, 1, ptr(dX), n, dX.shape[1], None, 0, 0, 8, 1.0, bp, noise, ptr(A), A.shape[1], 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = rt.lib.gpx_potrf(rt.h, n, ptr(A), A.shape[1], ptr(dinv))
    torch.cuda.synchronize()
    print("noise %5.2f potrf info %6d  %.3f ms" % (noise, info, 1e3 * (time.perf_counter() - t0)), flush=True)
gp = GaussianProcessRegression(ARD(sigma=1.0, beta=beta), sigma=0.1)
gp.fit(X, y)
for s in (0.1, 0.1000001, -0.5, 0.1000002, -0.4):
    gp.set_params(sigma=s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        gp.grad_log_marginal_likelihood()
        what = "ok"
    except np.linalg.LinAlgError as e:
        what = str(e)[:40]
    t1 = time.perf_counter() - t0
    gp.set_params(sigma=s + 1e-9)
    t0 = time.perf_counter()
    try:
        gp.log_marginal_likelihood()
    except np.linalg.LinAlgError:
        pass
    t2 = time.perf_counter() - t0
    print("sigma %10.7f lml+grad %.3f ms (%s)   lml only %.3f ms" % (s, 1e3 * t1, what, 1e3 * t2), flush=True)
