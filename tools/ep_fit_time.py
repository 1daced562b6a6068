"""Wall time of GaussianProcessExpectationPropagation.fit at n (make_moons, RBF) through the public API."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sklearn.datasets import make_moons
from gaussian_process_amd import RBF, GaussianProcessExpectationPropagation
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, y = make_moons(n, noise=0.1, random_state=0)
y = 2.0 * y - 1.0
for rep in range(2):
    np.random.seed(0)
    ep = GaussianProcessExpectationPropagation(RBF())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ep.fit(X, y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lml = ep.log_marginal_likelihood()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    acc = np.mean(ep.predict(X[:2000]) == y[:2000])
    torch.cuda.synchronize(); t3 = time.perf_counter()
print("EP fit() n=%d: %.3f s; log_marginal_likelihood %.3f s (%.6f); predict(2000) %.3f s (accuracy %.4f)"
      % (n, t1 - t0, t2 - t1, lml, t3 - t2, acc), flush=True)
