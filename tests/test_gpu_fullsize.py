"""Parity at BASELINE.json's full sizes, where neither the reference nor the oracle can run in test
time: size-independent properties of the results the HIP path returns (task rule 3).

  C4  regression n = 65536, d = 8: ||Ky alpha - y|| / ||y||, sampled entries of L L^T against Ky,
      the LML gradient against central differences of the LML, sanity of the predictive moments;
  C5  EP classification n = 8192 (make_moons): every site satisfies the moment-matching fixed point
      (expectation_propagation.py:76-85) with the converged tau, nu, diag(Sigma), mu.
"""
import ctypes as C

import numpy as np
import pytest
from scipy.special import log_ndtr

pytestmark = pytest.mark.gpu


def test_regression_n65536_properties():
    from gaussian_process_amd import ARD, GaussianProcessRegression, _lib
    from gaussian_process_amd.datasets import synthetic_ard
    from gaussian_process_amd.device import ptr
    n, m, noise = 65536, 512, 0.1
    X, y, Xs, beta = synthetic_ard(n, m=m, d=8, seed=0, noise=noise)
    gp = GaussianProcessRegression(ARD(sigma=1.0, beta=beta), sigma=noise)
    gp.fit(X, y)
    lml = gp.log_marginal_likelihood()
    assert np.isfinite(lml)
    alpha = gp.alpha_()
    dv = gp._dev
    rt = dv["rt"]
    rt.bind_stream()
    # (1) sampled entries of L L^T against Ky recomputed on the host from X
    L = dv["Ky"]
    worst = 0.0
    for i, j in [(0, 0), (n - 1, n - 1), (n - 1, 0), (n // 2 + 77, n // 2 - 5), (n - 3, n - 130), (n // 3, n // 7)]:
        got = float(rt.torch.dot(L[i, : j + 1], L[j, : j + 1]).item())
        ref = float(np.exp(-0.5 * np.sum(beta * (X[i] - X[j]) ** 2))) + (noise if i == j else 0.0)
        worst = max(worst, abs(got - ref))
    assert worst < 1e-12
    # (2) residual of the solve with a freshly built Ky (second 34 GB buffer; freed before the gradient)
    _, bp = _keep = _lib.host_doubles(beta)
    K2 = rt.matrix(n, n)
    _lib.check(rt.lib.gpx_gram(rt.h, 1, ptr(dv["X"]), n, dv["X"].shape[1], None, 0, 0, 8, 1.0, bp, noise, ptr(K2),
                               K2.shape[1], 0), "gpx_gram")
    a_d, out = rt.upload_vector(alpha), rt.vector(n)
    _lib.check(rt.lib.gpx_gemv(rt.h, 0, n, n, ptr(K2), K2.shape[1], ptr(a_d), ptr(out)), "gpx_gemv")
    res = rt.download(out[:n]) - y
    del K2
    assert np.linalg.norm(res) / np.linalg.norm(y) < 1e-10
    # (3) predictive moments: finite, variance in (0, prior variance + noise]
    mean, var = gp.predict_with_variance(Xs, full_cov=False)
    assert np.isfinite(mean).all() and np.all(var > 0.0) and np.all(var <= 1.0 + noise + 1e-12)
    # (4) gradient against central differences of the LML along the noise parameter
    g = gp.grad_log_marginal_likelihood()
    assert set(g) == {"sigma", "cov__sigma", "cov__beta"} and g["cov__beta"].shape == (8,)
    h = 1e-5   # truncation error of the central difference is 5e-4 (relative) at h = 1e-4, 5e-6 at 1e-5 here
    gp.set_params(sigma=noise + h)
    up_ = gp.log_marginal_likelihood()
    gp.set_params(sigma=noise - h)
    dn = gp.log_marginal_likelihood()
    assert abs((up_ - dn) / (2 * h) - g["sigma"][0]) < 1e-4 * abs(g["sigma"][0])


def test_ep_n8192_sites_are_at_the_moment_matching_fixed_point():
    from sklearn.datasets import make_moons
    from gaussian_process_amd import GaussianProcessExpectationPropagation, RBF
    n = 8192
    X, y = make_moons(n, noise=0.1, random_state=0)
    y = 2.0 * y - 1.0
    np.random.seed(0)
    ep = GaussianProcessExpectationPropagation(RBF())
    ep.fit(X, y)
    tau, nu, mu = ep.tau_tilde, ep.nu_tilde, ep.mu
    dS = ep._diag_Sigma()
    assert np.all(tau > 0) and np.all(dS > 0)
    # cavity and tilted moments of every site (expectation_propagation.py:76-82); at the fixed point the
    # tilted mean / variance equal the posterior marginal mu_i, Sigma_ii
    tau_cav = 1.0 / dS - tau
    nu_cav = mu / dS - nu
    dom = tau_cav ** 2 + tau_cav
    z = y * nu_cav / np.sqrt(dom)
    ratio = np.exp(-0.5 * z * z - 0.5 * np.log(2 * np.pi) - log_ndtr(z))
    mu_hat = nu_cav / tau_cav + y * ratio / np.sqrt(dom)
    var_hat = 1.0 / tau_cav - ratio / dom * (z + ratio)
    # the reference stops when the squared change of ALL site parameters in one sweep is < 1e-8, so each
    # site is within ~1e-4 of its fixed point
    assert np.max(np.abs(mu_hat - mu)) < 2e-3
    assert np.max(np.abs(var_hat - dS) / dS) < 2e-3
    assert np.mean(ep.predict(X[:4096]) == y[:4096]) > 0.99
    assert np.isfinite(ep.log_marginal_likelihood())
