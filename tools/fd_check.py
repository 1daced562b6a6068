"""Finite-difference check of d LML / d noise at n (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussian_process_amd import ARD, GaussianProcessRegression
from gaussian_process_amd.datasets import synthetic_ard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
noise = 0.1
X, y, Xs, beta = synthetic_ard(n, m=16, d=8, seed=0, noise=noise)
gp = GaussianProcessRegression(ARD(sigma=1.0, beta=beta), sigma=noise)
gp.fit(X, y)
g = gp.grad_log_marginal_likelihood()["sigma"][0]
print("n=%d analytic %.9f" % (n, g))
for h in (1e-3, 1e-4, 1e-5):
    gp.set_params(sigma=noise + h); up_ = gp.log_marginal_likelihood()
    gp.set_params(sigma=noise - h); dn = gp.log_marginal_likelihood()
    fd = (up_ - dn) / (2 * h)
    print("h=%g fd %.9f rel diff %.3e (LML up %.9f dn %.9f)" % (h, fd, abs(fd - g) / abs(g), up_, dn))
