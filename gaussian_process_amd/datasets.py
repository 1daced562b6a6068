"""Synthetic GP data (reference: gaussian_process/datasets.py:6-13).

``generate_gp`` keeps the reference's signature and draws X exactly as the reference does (same
generator calls, so X is identical for a given ``random_state``).  The function values are drawn as
``y = L z`` with ``L = chol(K + noise I)`` factored on the MI355X (gpx_gram + gpx_potrf + gpx_trmm)
instead of numpy's SVD-based ``multivariate_normal`` (datasets.py:12, O(n^3) on the host and the
reason the examples stop at n ~ 10^4): the same distribution, not the same numbers -- parity of the
sample VALUES with the reference is unpinned by construction (different factor, different use of the
random stream); what is tested is X, and that y has covariance L L^T for the z that was drawn.
"""
import ctypes as C

import numpy as np
from sklearn.utils import check_random_state

from . import _lib
from .device import on_device, ptr, runtime
from .kernels import RBF


@on_device
def generate_gp(n_samples=50, n_features=1, noise=0.01, range=(-3, 3),
                kernel=RBF(sigma=1.0, beta=1.0), random_state=None, return_latent=False):
    generator = check_random_state(random_state)
    X = generator.rand(n_samples, n_features) * (range[1] - range[0]) + range[0]
    z = generator.standard_normal(n_samples)
    kind, sigma_k, beta = kernel.gram_params(n_features)
    rt = runtime()
    rt.bind_stream()
    n = n_samples
    dX = rt.upload_matrix(X)
    Ky = rt.matrix(n, n)
    dinv = rt.dinv(n)
    _, beta_p = _keep = _lib.host_doubles(beta)
    ld = Ky.shape[1]
    _lib.check(rt.lib.gpx_gram(rt.h, kind, ptr(dX), n, dX.shape[1], None, 0, 0, n_features, float(sigma_k), beta_p,
                               float(noise), ptr(Ky), ld, _lib.GPX_LOWER), "gpx_gram")
    _lib.check(rt.lib.gpx_potrf(rt.h, n, ptr(Ky), ld, ptr(dinv)), "gpx_potrf")
    dz = rt.torch.zeros((n, 2), dtype=rt.torch.float64, device=rt.dev)
    dz[:, 0].copy_(rt.torch.from_numpy(z))
    out = rt.torch.empty((n, 2), dtype=rt.torch.float64, device=rt.dev)
    _lib.check(rt.lib.gpx_trmm(rt.h, 0, n, 1, ptr(Ky), ld, ptr(dz), 2, ptr(out), 2), "gpx_trmm")
    y = rt.download(out[:, 0]).copy()
    if return_latent:
        return X, y, z
    return X, y


def synthetic_ard(n, m=256, d=8, seed=0, noise=0.1, n_features=2048):
    """The benchmark workload of SURVEY.md section 8(d): X ~ U[-3, 3]^d (the reference's own range,
    datasets.py:6,10), true ARD beta* = 0.5 * logspace(-2, 0, d), y = a random-Fourier-feature draw
    from that GP (O(n); the reference's multivariate_normal is O(n^3)) + N(0, noise).  Host NumPy:
    it only makes inputs.  Returns (X, y, Xs, beta*)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3.0, 3.0, size=(n, d))
    Xs = rng.uniform(-3.0, 3.0, size=(m, d))
    beta = 0.5 * np.logspace(-2.0, 0.0, d)
    W = rng.standard_normal((d, n_features)) * np.sqrt(beta)[:, None]
    b = rng.uniform(0.0, 2.0 * np.pi, size=n_features)
    w = rng.standard_normal(n_features)
    f = np.sqrt(2.0 / n_features) * (np.cos(X @ W + b) @ w)
    y = f + np.sqrt(noise) * rng.standard_normal(n)
    return X, y, Xs, beta
