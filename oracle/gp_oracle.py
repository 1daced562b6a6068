"""CPU oracle for the exact-GP hot path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

This module is a NumPy/SciPy restatement of the arithmetic of the reference
``emakryo/gaussian_process`` hot path.  It exists only so that ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg can check
and time the HIP path against it.  Nothing under ``gaussian_process_amd/`` may
import it: the product path has no CPU fallback.

Pinning: every function here is checked against outputs of the real reference
(imported from ``/root/reference`` in the build container by
``tests/golden/make_golden.py``) through the committed fixtures in
``tests/golden/*.npz`` -- see ``tests/test_oracle_golden.py``.

Reference anchors (file:line in /root/reference/gaussian_process):
  rbf_cov            kernels.py:21-46
  ard_cov            kernels.py:61-88
  gpr_predict        regression.py:31-44
  gpr_lml            regression.py:46-55
  gpr_grad           regression.py:57-76
  ep_fit / ep_*      expectation_propagation.py:25-167
  distillation_*     distillation.py:22-63

Differences from the reference that do not change results beyond rounding:
  * the ARD Gram is accumulated one input dimension at a time and in row
    blocks, so the (n, n, d) temporaries of kernels.py:77-80 never exist and
    the oracle can run at n = 16384+ in bounded memory;
  * the gradient contracts A with dK/dtheta one dimension at a time instead of
    materialising the (d, n, n) stack (regression.py:63-72);
  * Ky^-1 is formed from the Cholesky factor (potri) when ``spd_inverse=True``
    instead of the LU inverse of regression.py:60 (default follows the
    reference: scipy.linalg.inv).
Quirks kept on purpose: the noise term is ``sigma * I`` (a variance), the
kernel amplitude is ``sigma`` (not squared), the predictive covariance is the
full m x m matrix with ``+ sigma * I`` (regression.py:41-44).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
from scipy.stats import norm

EP_EPS = 1e-8  # expectation_propagation.py:7


# --------------------------------------------------------------------------
# covariance functions
# --------------------------------------------------------------------------
def rbf_cov(X1, X2=None, sigma=1.0, beta=1.0, want_grad=True):
    """kernels.py:34-39 -- isotropic RBF through the |x|^2+|y|^2-2xy expansion.

    Returns (K, dsigma, dbeta); the derivatives are None when want_grad=False.
    """
    X1 = np.asarray(X1, dtype=np.float64)
    X2 = X1 if X2 is None else np.asarray(X2, dtype=np.float64)
    sq1 = np.einsum("ij,ij->i", X1, X1)[:, None]
    sq2 = np.einsum("ij,ij->i", X2, X2)[None, :]
    r2 = sq1 + sq2 - 2.0 * (X1 @ X2.T)
    K = sigma * np.exp(-0.5 * beta * r2)
    if not want_grad:
        return K, None, None
    return K, K / sigma, -0.5 * r2 * K


def ard_sqdist_dim(X1, X2, k):
    """(x_ik - x'_jk)^2 for one input dimension k (kernels.py:77, one slice)."""
    dk = X1[:, k][:, None] - X2[:, k][None, :]
    return dk * dk


def ard_cov(X1, X2=None, sigma=1.0, beta=None, want_grad=False, row_block=2048):
    """kernels.py:77-81 -- ARD kernel, blockwise, one dimension at a time.

    Returns (K, dsigma, dbeta) with dbeta of shape (d, n1, n2) only when
    want_grad=True (small problems / golden checks).
    """
    X1 = np.asarray(X1, dtype=np.float64)
    X2 = X1 if X2 is None else np.asarray(X2, dtype=np.float64)
    beta = np.asarray(beta, dtype=np.float64)
    n1, d = X1.shape
    n2 = X2.shape[0]
    K = np.empty((n1, n2))
    for r0 in range(0, n1, row_block):
        r1 = min(n1, r0 + row_block)
        acc = np.zeros((r1 - r0, n2))
        for k in range(d):
            acc += beta[k] * ard_sqdist_dim(X1[r0:r1], X2, k)
        K[r0:r1] = sigma * np.exp(-0.5 * acc)
    if not want_grad:
        return K, None, None
    dbeta = np.stack([-0.5 * ard_sqdist_dim(X1, X2, k) * K for k in range(d)])
    return K, K / sigma, dbeta


def cov(kind, X1, X2=None, sigma=1.0, beta=1.0, want_grad=False):
    if kind == "rbf":
        return rbf_cov(X1, X2, sigma, float(np.asarray(beta).reshape(-1)[0]), want_grad)
    if kind == "ard":
        return ard_cov(X1, X2, sigma, beta, want_grad)
    raise ValueError(kind)


# --------------------------------------------------------------------------
# exact GP regression
# --------------------------------------------------------------------------
def gpr_factor(kind, X, sigma_k, beta, noise):
    """Ky = K + noise*I and its lower Cholesky factor (regression.py:47-49)."""
    K, _, _ = cov(kind, X, None, sigma_k, beta, want_grad=False)
    K[np.diag_indices_from(K)] += noise
    L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    return L


def gpr_lml(kind, X, y, sigma_k, beta, noise, L=None):
    """regression.py:46-55.  Returns (lml, a = L^-1 y, L)."""
    n = X.shape[0]
    if L is None:
        L = gpr_factor(kind, X, sigma_k, beta, noise)
    a = sla.solve_triangular(L, y, lower=True, check_finite=False)
    lml = -0.5 * a.dot(a) - np.log(np.diagonal(L)).sum() - 0.5 * n * np.log(2.0 * np.pi)
    return float(lml), a, L


def gpr_alpha(L, y):
    """alpha = Ky^-1 y (regression.py:61) from the factor."""
    a = sla.solve_triangular(L, y, lower=True, check_finite=False)
    return sla.solve_triangular(L, a, lower=True, trans="T", check_finite=False)


def gpr_predict(kind, Xtr, ytr, Xte, sigma_k, beta, noise, L=None, full_cov=True):
    """regression.py:31-44.  mean (m,), cov (m,m) incl. + noise*I (or its diagonal)."""
    if L is None:
        L = gpr_factor(kind, Xtr, sigma_k, beta, noise)
    a = sla.solve_triangular(L, ytr, lower=True, check_finite=False)
    ks, _, _ = cov(kind, Xtr, Xte, sigma_k, beta, want_grad=False)
    V = sla.solve_triangular(L, ks, lower=True, check_finite=False)
    mean = a @ V
    if full_cov:
        kss, _, _ = cov(kind, Xte, None, sigma_k, beta, want_grad=False)
        c = kss - V.T @ V
        c[np.diag_indices_from(c)] += noise
        return mean, c
    # diagonal only: k(x,x) = sigma_k for both kernels up to rounding
    if kind == "rbf":
        kss = np.diagonal(rbf_cov(Xte, None, sigma_k, float(np.asarray(beta).reshape(-1)[0]), False)[0]) \
            if Xte.shape[0] <= 4096 else np.full(Xte.shape[0], sigma_k)
    else:
        kss = np.full(Xte.shape[0], sigma_k)
    return mean, kss - np.einsum("ij,ij->j", V, V) + noise


def gpr_grad(kind, X, y, sigma_k, beta, noise, spd_inverse=False, L=None):
    """regression.py:57-76.  Dict in the reference key order sigma, cov__sigma, cov__beta."""
    X = np.asarray(X, dtype=np.float64)
    n, d = X.shape
    if kind == "rbf":
        K, dsig, dbeta = rbf_cov(X, None, sigma_k, float(np.asarray(beta).reshape(-1)[0]), True)
    else:
        K, _, _ = ard_cov(X, None, sigma_k, beta, False)
    Ky = K.copy()
    Ky[np.diag_indices_from(Ky)] += noise
    if spd_inverse:
        if L is None:
            L = sla.cholesky(Ky, lower=True, check_finite=False)
        Kinv = sla.cho_solve((L, True), np.eye(n), check_finite=False)
        alpha = sla.cho_solve((L, True), y, check_finite=False)
    else:
        Kinv = sla.inv(Ky, check_finite=False)
        alpha = sla.solve(Ky, y, assume_a="pos", check_finite=False)
    A = np.outer(alpha, alpha) - Kinv
    out = {"sigma": np.array([0.5 * np.trace(A)]),
           "cov__sigma": np.array([0.5 * np.sum(A * (K / sigma_k))])}
    if kind == "rbf":
        out["cov__beta"] = np.array([0.5 * np.sum(A * dbeta)])
    else:
        AK = A * K
        out["cov__beta"] = np.array(
            [0.5 * np.sum(AK * (-0.5 * ard_sqdist_dim(X, X, k))) for k in range(d)])
    return out, alpha


# --------------------------------------------------------------------------
# expectation propagation (probit)  -- expectation_propagation.py:25-108
# --------------------------------------------------------------------------
def ep_recompute(K, tau, nu):
    """expectation_propagation.py:61-65 / 90-94 -- the linear algebra the GPU path reroutes."""
    n = K.shape[0]
    s = np.sqrt(tau)
    B = np.eye(n) + (s[:, None] * K) * s[None, :]
    L = sla.cholesky(B, lower=True, check_finite=False)
    V = sla.solve_triangular(L, s[:, None] * K, lower=True, check_finite=False)
    Sigma = K - V.T @ V
    mu = Sigma @ nu
    return L, Sigma, mu, s


def ep_site_update(i, yi, Sigma, mu, tau, nu):
    """One site of the Gauss-Seidel sweep, expectation_propagation.py:76-88 (in place)."""
    sii = Sigma[i, i]
    tau_cav = 1.0 / sii - tau[i]
    nu_cav = mu[i] / sii - nu[i]
    dom = tau_cav * tau_cav + tau_cav
    z = yi * nu_cav / np.sqrt(dom)
    ratio = np.exp(norm.logpdf(z) - norm.logcdf(z))
    mu_hat = nu_cav / tau_cav + yi * ratio / np.sqrt(dom)
    var_hat = 1.0 / tau_cav - ratio / dom * (z + ratio)
    dtau = 1.0 / var_hat - tau_cav - tau[i]
    tau[i] += dtau
    nu[i] = mu_hat / var_hat - nu_cav
    col = Sigma[i].copy()
    Sigma -= (dtau / (1.0 + dtau * sii)) * np.outer(col, col)
    mu[:] = Sigma @ nu


def ep_fit(K, y, max_sweeps=100, orders=None, fix_index=None, nu_fixed=None, tau_fixed=None):
    """expectation_propagation.py:25-108 with shuffle=False (or explicit site orders).

    ``fix_index`` / ``nu_fixed`` / ``tau_fixed``: sites whose natural parameters are given and never
    updated (expectation_propagation.py:26-51, 73-74; used by distillation.py:33-35).
    Returns a dict with tau, nu, mu, Sigma, K (jittered), L, S, z, sweeps.
    """
    n = K.shape[0]
    K = K + EP_EPS * np.eye(n)
    tau = np.zeros(n)
    nu = np.zeros(n)
    frozen = np.zeros(n, dtype=bool)
    if fix_index is not None:
        fix_index = np.asarray(fix_index, dtype=int)
        frozen[fix_index] = True
        tau[fix_index] = np.asarray(tau_fixed, dtype=float)
        nu[fix_index] = np.asarray(nu_fixed, dtype=float)
    L, Sigma, mu, s = ep_recompute(K, tau, nu)
    sweeps = 0
    for c in range(max_sweeps):
        tau_old = tau.copy()
        nu_old = nu.copy()
        order = np.arange(n) if orders is None else orders[c]
        for i in order:
            if frozen[i]:
                continue
            ep_site_update(int(i), y[i], Sigma, mu, tau, nu)
        L, Sigma, mu, s = ep_recompute(K, tau, nu)
        sweeps = c + 1
        if np.sum((tau_old - tau) ** 2) < EP_EPS and np.sum((nu_old - nu) ** 2) < EP_EPS:
            break
    w = sla.solve_triangular(L, s * (K @ nu), lower=True, check_finite=False)
    z = s * sla.solve_triangular(L, w, lower=True, trans="T", check_finite=False)
    return dict(tau=tau, nu=nu, mu=mu, Sigma=Sigma, K=K, L=L, S=s, z=z, sweeps=sweeps)


def ep_posterior(state, Ks, Kss):
    """expectation_propagation.py:115-119."""
    mean = Ks.T @ (state["nu"] - state["z"])
    v = sla.solve_triangular(state["L"], state["S"][:, None] * Ks, lower=True, check_finite=False)
    return mean, Kss - np.sum(v * v, axis=0)


def ep_lml(state, y):
    """expectation_propagation.py:129-150 (sum of t0..t6)."""
    Sigma, mu, tau, nu, K, L, s = (state[k] for k in ("Sigma", "mu", "tau", "nu", "K", "L", "S"))
    dS = np.diag(Sigma)
    tau_cav = 1.0 / dS - tau
    nu_cav = mu / dS - nu
    t0 = -np.sum(np.log(np.diag(L)))
    t1 = 0.5 * np.sum(np.log(1.0 + tau / tau_cav))
    Knu = K @ nu
    t2 = 0.5 * nu.dot(Knu)
    w = sla.solve_triangular(L, s * Knu, lower=True, check_finite=False)
    t3 = -0.5 * w.dot(w)
    t4 = -0.5 * (nu ** 2).dot(1.0 / (tau_cav + tau))
    mu_cav = nu_cav / tau_cav
    t5 = 0.5 * nu_cav.dot((tau * mu_cav - 2.0 * nu) / (tau_cav + tau))
    t6 = np.sum(norm.logcdf(y * mu_cav / np.sqrt(1.0 + 1.0 / tau_cav)))
    return float(t0 + t1 + t2 + t3 + t4 + t5 + t6)


def ep_grad(state, dK):
    """expectation_propagation.py:161-166 for ONE derivative matrix dK (n x n): the pair
    [b^T dK b / 2, tr(U^T U dK) / 2] with U = L^-1 diag(S), b = nu - U^T U K nu (K jittered, as :163).
    For ARD the caller passes the planes of kernels.py:80 one at a time (the reference's np.trace over
    the whole (d, n, n) stack is not a per-dimension gradient)."""
    L, s, K, nu = state["L"], state["S"], state["K"], state["nu"]
    U = sla.solve_triangular(L, np.diag(s), lower=True, check_finite=False)
    UtU = U.T @ U
    b = nu - UtU @ (K @ nu)
    return [0.5 * float(b @ dK @ b), 0.5 * float(np.sum(UtU * dK.T))]


# --------------------------------------------------------------------------
# distillation: EP over a two-task Kronecker kernel -- distillation.py:22-63
# --------------------------------------------------------------------------
def distillation_kernel(K_, rho):
    """distillation.py:27-32: kron([[1, rho], [rho, 1]], K_)."""
    return np.kron(np.array([[1.0, rho], [rho, 1.0]]), K_)


def distillation_fit(K_, y, s, rho, sigma, orders=None, max_sweeps=100):
    """distillation.py:22-35: the n soft-label sites (indices n..2n-1) are fixed at
    nu = s / sigma, tau = 1 / sigma; the n classification sites (0..n-1) are swept."""
    n = K_.shape[0]
    return ep_fit(distillation_kernel(K_, rho), y, max_sweeps=max_sweeps, orders=orders,
                  fix_index=np.arange(n, 2 * n), nu_fixed=np.asarray(s, dtype=float) / sigma,
                  tau_fixed=np.ones(n) / sigma)


def distillation_posterior(state, Ks_, Kss, rho):
    """distillation.py:40-44."""
    return ep_posterior(state, np.vstack([Ks_, rho * Ks_]), Kss)


def distillation_lml(state, y, tol=1e-7):
    """distillation.py:46-63 as written, including its use of trace(L) in t0 and of the
    soft-label half of the cavity parameters in t3 (SURVEY.md marks the formula as suspect;
    the port keeps it so results stay comparable)."""
    Sigma, mu, tau, nu, L, S = (state[k] for k in ("Sigma", "mu", "tau", "nu", "L", "S"))
    n = len(y)
    dS = np.diag(Sigma)
    tau_cav = 1.0 / dS - tau
    nu_cav = mu / dS - nu
    t0 = -np.log(np.trace(L))
    t1 = np.sum(np.log(S))
    w = np.linalg.solve(L @ np.diag(S), nu)
    t2 = -0.5 * w.dot(w)
    t3 = np.sum(norm.logcdf(y * nu_cav[n:] / np.sqrt(tau_cav[n:] * (1.0 + tau_cav[n:]))))
    assert np.all(tau_cav[n:] > tol)
    assert np.all(tau[n:] > tol)
    t4 = 0.5 * np.sum(np.log((1.0 / tau_cav + 1.0 / tau)[n:]))
    t5 = 0.5 * np.sum(((nu_cav / tau_cav + nu / tau) ** 2 * (1.0 / tau_cav + 1.0 / tau))[n:])
    return float(t0 + t1 + t2 + t3 + t4 + t5)


# --------------------------------------------------------------------------
# synthetic workload of SURVEY.md section 8(d) / BASELINE.md section 3
# --------------------------------------------------------------------------
def synthetic_ard(n, m=256, d=8, seed=0, noise=0.1, n_features=2048):
    """X ~ U[-3,3]^d, beta* = 0.5*logspace(-2,0,d), y = RFF draw of that GP + N(0, noise)."""
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
