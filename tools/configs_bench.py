"""The BASELINE configs other than the headline (development tool; numbers quoted in DESIGN.md):
C2  2-D ARD regression with LML hyper-parameter optimisation (ARD_regression_2d flow) at the notebook's
    size (goldens) and scaled up to n = 4096, counting optimiser evaluations;
C3  synthetic ARD regression n = 16384, d = 8: one fused LML+grad evaluation + one prediction."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from gaussian_process_amd import ARD, GaussianProcessRegression
from gaussian_process_amd.datasets import synthetic_ard


class Counting(GaussianProcessRegression):
    evals = 0

    def _evaluate(self, need_grad):
        st = getattr(self, "_state", None)
        hit = st is not None and st["key"] == self._theta_key() and (st["grad"] is not None or not need_grad)
        if not hit:
            Counting.evals += 1
        return super()._evaluate(need_grad)


def c2(n):
    rng = np.random.default_rng(0)
    X = rng.uniform(-3, 3, (n, 2))
    y = np.sin(X[:, 0]) * np.cos(2.0 * X[:, 1]) + 0.1 * rng.standard_normal(n)
    Xs = rng.uniform(-3, 3, (1000, 2))
    gp = Counting(cov=ARD(beta=np.array([1.0, 1.0])))
    gp.fit(X, y)
    gp.log_marginal_likelihood()          # warm-up (library load, first launches)
    Counting.evals = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp.fit(X, y, empirical_bayes=True)
    mean = gp.predict(Xs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("C2 n=%d: empirical_bayes + predict %.3f s, %d fused evaluations (%.1f ms each), beta -> %s, LML %.4f"
          % (n, dt, Counting.evals, 1e3 * dt / max(Counting.evals, 1), np.round(gp.cov.beta, 4), gp.log_marginal_likelihood()),
          flush=True)


def c3(n=16384, m=4096):
    X, y, Xs, beta = synthetic_ard(n, m=m, d=8, seed=0, noise=0.1)
    gp = GaussianProcessRegression(ARD(sigma=1.0, beta=beta), sigma=0.1)
    gp.fit(X, y)
    for rep in range(2):
        gp._state = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lml = gp.log_marginal_likelihood()      # LML first, as base.py:35-37: the gradient goes on from its factor
        g = gp.grad_log_marginal_likelihood()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        mean, cov = gp.predict_with_variance(Xs)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print("C3 n=%d m=%d: fused LML+grad %.3f s (%.1f TFLOP/s on n^3 flops), predict after the gradient (full cov, from the inverse factor, incl. download) %.3f s, LML %.4f"
          % (n, m, t1 - t0, n ** 3 / (t1 - t0) / 1e12, t2 - t1, lml), flush=True)


if __name__ == "__main__":
    c2(400)
    c2(4096)
    c3()
