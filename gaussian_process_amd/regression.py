"""Exact GP regression on the MI355X (reference: gaussian_process/regression.py).

Same constructor, methods, attribute names, parameter names/bounds and return shapes as the
reference class (regression.py:8-76); the arithmetic runs in libgpx:

  predict_with_variance   regression.py:31-44  -> gpx_gpr_lml_grad (factor) + gpx_gpr_predict, or
                                                  gpx_gpr_predict_inv after a gradient evaluation
  log_marginal_likelihood regression.py:46-55  -> gpx_gpr_lml_grad (grad_host = NULL)
  grad_log_marginal_...   regression.py:57-76  -> gpx_gpr_lml_grad (fused LML + gradient), or
                                                  gpx_gpr_grad_from_factor after an LML evaluation

Quirks kept: the noise enters as ``sigma * I`` (a variance); ``fit`` returns None; the
predictive covariance is the full m x m matrix including ``+ sigma * I``; the gradient dict is
keyed ``sigma, cov__sigma, cov__beta`` with shapes (1,), (1,), (1,)|(d,).

Unlike the reference nothing is recomputed needlessly: the training set stays resident in HBM
after ``fit`` and one factorisation is cached per (data, hyper-parameter) state.
"""
import ctypes as C

import numpy as np
from sklearn.base import RegressorMixin
from sklearn.utils import check_X_y

from . import _lib
from .base import BayesEstimator
from .device import on_device, ptr, runtime


class GaussianProcessRegression(RegressorMixin, BayesEstimator):
    """Ordinary Gaussian process regression; drop-in for the reference estimator."""

    def __init__(self, cov, sigma=0.01):
        self.cov = cov
        self.sigma = sigma
        self.param_bounds = {"sigma": (1e-5, None)}
        for name, bound in cov.param_bounds.items():
            self.param_bounds["cov__" + name] = bound

    # ------------------------------------------------------------------ data
    def fit(self, X, y, empirical_bayes=False):
        X, y = check_X_y(X, y, y_numeric=True)
        self.Xtr = np.ascontiguousarray(X, dtype=np.float64)
        self.ytr = np.ascontiguousarray(y, dtype=np.float64)
        self.n, self.dim = self.Xtr.shape
        self._dev = None     # device copies of the data
        self._state = None   # cached factorisation / evaluation
        if empirical_bayes:
            self.empirical_bayes()

    def refit(self):
        # Nothing depends on the hyper-parameters at fit time (regression.py:25-26): keep the
        # device-resident data, drop only the cached evaluation if theta moved (checked lazily).
        if getattr(self, "_dev", None) is None:
            self.fit(self.Xtr, self.ytr, empirical_bayes=False)

    # ------------------------------------------------------------------ device state
    def _theta_key(self):
        kind, sigma_k, beta = self.cov.gram_params(self.dim)
        return (kind, float(sigma_k), tuple(np.asarray(beta, dtype=np.float64).tolist()),
                float(self.sigma))

    def _device_data(self):
        if getattr(self, "_dev", None) is None:
            rt = runtime()
            n = self.n
            self._dev = dict(rt=rt, X=rt.upload_matrix(self.Xtr), y=rt.upload_vector(self.ytr),
                             Ky=rt.matrix(n, n), dinv=rt.dinv(n), vec=rt.vector(5 * n),
                             alpha=rt.vector(n), a=rt.vector(n))
        return self._dev

    @on_device
    def _evaluate(self, need_grad):
        """One fused evaluation at the current theta; cached until theta or the data change."""
        key = self._theta_key()
        st = getattr(self, "_state", None)
        if st is not None and st["key"] == key and (st["grad"] is not None or not need_grad):
            return st
        dv = self._device_data()
        rt = dv["rt"]
        rt.bind_stream()
        n = self.n
        kind, sigma_k, beta = key[0], key[1], np.array(key[2])
        # the call overwrites the resident factor: whatever was cached is gone from here on, also if
        # the call raises (LinAlgError at a non-PD trial theta of L-BFGS-B)
        self._state = None
        _, beta_p = _keep = _lib.host_doubles(beta)
        lml = C.c_double(0.0)
        ngrad = 2 + len(beta)
        grad = (C.c_double * ngrad)()
        ld = dv["Ky"].shape[1]
        if st is not None and st["key"] == key and st["has_factor"]:
            # LML first, gradient second at the same theta (the order of base.py:35-37): go on from the
            # factor and alpha of the LML evaluation, no second Gram matrix or factorisation
            rc = rt.lib.gpx_gpr_grad_from_factor(
                rt.h, kind, ptr(dv["X"]), n, dv["X"].shape[1], self.dim, sigma_k, beta_p, ptr(dv["Ky"]), ld,
                ptr(dv["dinv"]), ptr(dv["vec"]), ptr(dv["alpha"]), C.cast(grad, C.POINTER(C.c_double)))
            _lib.check(rc, "gpx_gpr_grad_from_factor")
            self._state = dict(key=key, lml=st["lml"], grad=np.array(list(grad), dtype=np.float64), has_factor=False)
            return self._state
        rc = rt.lib.gpx_gpr_lml_grad(
            rt.h, kind, ptr(dv["X"]), n, dv["X"].shape[1], self.dim, sigma_k, beta_p, key[3],
            ptr(dv["y"]), ptr(dv["Ky"]), ld, ptr(dv["dinv"]),
            ptr(dv["vec"]), ptr(dv["alpha"]), ptr(dv["a"]), C.byref(lml),
            C.cast(grad, C.POINTER(C.c_double)) if need_grad else None)
        _lib.check(rc, "gpx_gpr_lml_grad")
        g = None
        if need_grad:
            g = np.array(list(grad), dtype=np.float64)
        self._state = dict(key=key, lml=float(lml.value), grad=g, has_factor=not need_grad)
        return self._state

    # ------------------------------------------------------------------ reference API
    def log_marginal_likelihood(self):
        return self._evaluate(need_grad=False)["lml"]

    def grad_log_marginal_likelihood(self):
        g = self._evaluate(need_grad=True)["grad"]
        return {"sigma": np.array([g[0]]), "cov__sigma": np.array([g[1]]), "cov__beta": g[2:].copy()}

    @on_device
    def alpha_(self):
        """alpha = Ky^-1 y (regression.py:61) of the cached evaluation, as a NumPy array."""
        self._evaluate(need_grad=False)
        dv = self._dev
        return dv["rt"].download(dv["alpha"][: self.n])

    def predict(self, X):
        """Predictive mean (regression.py:28-29).  Only the mean is returned, so the m x m covariance
        the reference builds and discards is never formed, and the test points are streamed through
        the device in batches."""
        return self.predict_with_variance(X, full_cov=False)[0]

    def _predict_buffers(self, m, full_cov, need_Ks):
        """Prediction workspaces (V and, after a gradient pass, k*: n x m each; the m x m covariance), kept
        with the estimator and reused by later calls with the same number of test points -- a streamed
        prediction or an optimiser loop does not allocate per call (2 x 2.1 GB at n = 65536, m = 4096)."""
        dv = self._dev
        rt = dv["rt"]
        pb = dv.get("pred")
        if pb is None or pb["m"] != m:
            pb = dv["pred"] = dict(m=m, V=rt.matrix(self.n, m), Ks=None, mean=rt.vector(m), cov=None, var=rt.vector(m))
        if need_Ks and pb["Ks"] is None:
            pb["Ks"] = rt.matrix(self.n, m)
        if full_cov and pb["cov"] is None:
            pb["cov"] = rt.matrix(m, m)
        return pb["V"], pb["Ks"], pb["mean"], pb["cov"] if full_cov else None, None if full_cov else pb["var"]

    @on_device
    def predict_with_variance(self, X, full_cov=True, batch=16384):
        """(mean (m,), cov (m, m)) as regression.py:31-44; ``full_cov=False`` returns the
        diagonal variance (m,) instead, never forms the m x m matrix and tiles over the test points
        ``batch`` at a time (device memory n x batch, any m)."""
        X = np.ascontiguousarray(np.asarray(X), dtype=np.float64)
        assert X.ndim == 2 and X.shape[1] == self.dim
        if not np.isfinite(X).all():
            # regression.py:38: scipy's solve_triangular (check_finite=True) raises on the NaN column of k*
            raise ValueError("array must not contain infs or NaNs")
        m = X.shape[0]
        if not full_cov and m > batch:
            parts = [self.predict_with_variance(X[i:i + batch], full_cov=False, batch=batch)
                     for i in range(0, m, batch)]
            return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        st = getattr(self, "_state", None)
        if st is None or st["key"] != self._theta_key():
            st = self._evaluate(need_grad=False)
        dv = self._dev
        rt = dv["rt"]
        rt.bind_stream()
        key = st["key"]
        _, beta_p = _keep = _lib.host_doubles(np.array(key[2]))
        dXs = rt.upload_matrix(X)
        V, Ks, mean, cov, var = self._predict_buffers(m, full_cov, need_Ks=not st["has_factor"])
        if st["has_factor"]:
            rc = rt.lib.gpx_gpr_predict(
                rt.h, key[0], ptr(dv["X"]), self.n, dv["X"].shape[1], self.dim, key[1], beta_p, key[3],
                None,      # a = L^-1 y is still in vec[2n, 3n) from the LML evaluation that left this factor
                ptr(dv["Ky"]), dv["Ky"].shape[1], ptr(dv["dinv"]), ptr(dXs), m, dXs.shape[1],
                ptr(V), V.shape[1], ptr(dv["vec"]), ptr(mean), ptr(cov) if full_cov else None,
                cov.shape[1] if full_cov else 0, None if full_cov else ptr(var), 1 if full_cov else 0)
            _lib.check(rc, "gpx_gpr_predict")
        else:
            # the gradient pass left W = L^-1 in Ky and a = L^-1 y in dv["a"]: V = W k*, no refactorisation
            rc = rt.lib.gpx_gpr_predict_inv(
                rt.h, key[0], ptr(dv["X"]), self.n, dv["X"].shape[1], self.dim, key[1], beta_p, key[3],
                ptr(dv["a"]), ptr(dv["Ky"]), dv["Ky"].shape[1], ptr(dXs), m, dXs.shape[1], ptr(Ks),
                ptr(V), V.shape[1], ptr(mean), ptr(cov) if full_cov else None,
                cov.shape[1] if full_cov else 0, None if full_cov else ptr(var), 1 if full_cov else 0)
            _lib.check(rc, "gpx_gpr_predict_inv")
        mean_h = rt.download(mean[:m])
        if full_cov:
            return mean_h, rt.download(cov[:m, :m])
        return mean_h, rt.download(var[:m])
