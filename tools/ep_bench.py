"""BASELINE configs[4]: EP binary classification, make_moons(n), RBF: time per sweep split into the
device site loop and the recompute (development tool)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sklearn.datasets import make_moons
from gaussian_process_amd import RBF, GaussianProcessExpectationPropagation

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, y = make_moons(n, noise=0.1, random_state=0)
y = 2.0 * y - 1.0
np.random.seed(0)
gp = GaussianProcessExpectationPropagation(RBF())
gp.Xtr, gp.ytr, gp.n, gp.dim = X, y, n, 2
K = gp.cov(X).K
# instrument: time sweep and recompute separately
orig_sweep, orig_rec = gp._sweep, gp._recompute
tim = {"sweep": [], "recompute": []}
def timed(name, f):
    def g(*a):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a); torch.cuda.synchronize()
        tim[name].append(time.perf_counter() - t0); return r
    return g
gp._sweep, gp._recompute = timed("sweep", orig_sweep), timed("recompute", orig_rec)
t0 = time.perf_counter()
gp._ep(K)
tot = time.perf_counter() - t0
print("EP n=%d: %d sweeps, total %.2f s; site sweep %.3f s each (%.1f us/site; the reference's rank-1 form would move %.0f GB per sweep); recompute %.3f s each"
      % (n, len(tim["sweep"]), tot, np.mean(tim["sweep"]), 1e6 * np.mean(tim["sweep"]) / n,
         16.0 * n * n * n / 1e9, np.mean(tim["recompute"][1:])), flush=True)
print("train accuracy %.4f  LML %.6f" % (np.mean(gp.predict(X[:2000]) == y[:2000]), gp.log_marginal_likelihood()), flush=True)
