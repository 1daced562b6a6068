"""Covariance objects with the reference's interface (reference: gaussian_process/kernels.py).

``RBF(sigma, beta)`` and ``ARD(sigma, beta[d])`` are callables returning
``KernelValue(K, dtheta)`` exactly as kernels.py:21-46 / 61-88 do, but the matrices are built
by the HIP Gram kernels of libgpx (``gpx_gram`` / ``gpx_gram_dtheta``).  The estimators do not
go through ``__call__`` on their hot path: they hand ``kind / sigma / beta`` to the fused C
entry points, so the (d, n, n) derivative stack only exists when a user asks for ``dtheta``.

Deliberate differences:
  * inputs are up-cast to float64 (the reference keeps the incoming dtype);
  * ``ARD(...)(X, diag=True)`` returns ``(sigma*1, zeros((d, n)), ones)``; the reference
    raises TypeError at kernels.py:84 (bug);
  * RBF is evaluated from exact differences, not the |x|^2+|y|^2-2xy expansion
    (kernels.py:34-36); the two agree to ~2e-15 and the diagonal is exactly ``sigma``.
"""
from collections import namedtuple
from collections.abc import Mapping

import numpy as np
from sklearn.base import BaseEstimator

from . import _lib
from .device import on_device, ptr, runtime

KernelValue = namedtuple("KernelValue", ["K", "dtheta"])


class LazyDtheta(Mapping):
    """``KernelValue.dtheta`` of the device kernels: the mapping {'sigma': dK/dsigma, 'beta': dK/dbeta} of
    kernels.py:45,87 (same keys, same insertion order, same dense arrays), but the derivative planes -- for ARD a
    (d, n1, n2) stack, kernels.py:80 -- are built on the device and downloaded only when an entry is first READ.
    Callers that want K alone (the reference computes and throws the planes away at regression.py:32-33,41 and
    expectation_propagation.py:19,111) never pay for them; the estimators' gradients contract them on the device
    without materialising them at all (gpx_grad_trace)."""
    _KEYS = ("sigma", "beta")

    def __init__(self, build):
        self._build, self._planes = build, None

    def _get(self):
        if self._planes is None:
            self._planes = self._build()
            self._build = None
        return self._planes

    def __getitem__(self, key):
        if key not in self._KEYS:
            raise KeyError(key)
        return self._get()[key]

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)

    def __repr__(self):
        return "LazyDtheta(%s)" % ("materialised" if self._planes is not None else "pending")

_LOWER_BOUND = 1e-5


def _as_2d(X, name):
    X = np.asarray(X)
    assert X.ndim == 2, "%s must be 2-dimensional" % name
    return np.ascontiguousarray(X, dtype=np.float64)


def _nan_rows(X):
    """Rows of X with a non-finite coordinate (None if there are none).  The device exp clamps its argument
    with a v_min_f64, which drops NaNs (csrc/gram.hip: exp_neg_tab); NumPy's exp in kernels.py:37,79 propagates
    them, so the entries of those rows / columns are put back to NaN on the host."""
    bad = ~np.isfinite(X).all(axis=1)
    return bad if bad.any() else None


def _restore_nan(arr, bad1, bad2):
    if arr is None:
        return arr
    if bad1 is not None:
        arr[..., bad1, :] = np.nan
    if bad2 is not None:
        arr[..., :, bad2] = np.nan
    return arr


def clone_params(kernel):
    """A copy of a kernel object with its current parameter values (arrays copied)."""
    import copy
    return copy.deepcopy(kernel)


class _StationaryKernel(BaseEstimator):
    """Shared device path of RBF and ARD."""

    kind = None

    def _beta_vector(self, dim):
        raise NotImplementedError

    def gram_params(self, dim):
        """(kind id, sigma, beta array) as the C ABI wants them."""
        return _lib.GPX_KIND[self.kind], float(self.sigma), self._beta_vector(dim)

    @on_device
    def _evaluate(self, X1, X2, want_dtheta=True):
        rt = runtime()
        rt.bind_stream()
        lib = rt.lib
        n1, dim = X1.shape
        n2 = X2.shape[0]
        kind, sigma, beta = self.gram_params(dim)
        beta_arr, beta_p = _lib.host_doubles(beta)
        dX1 = rt.upload_matrix(X1)
        dX2 = dX1 if X2 is X1 else rt.upload_matrix(X2)
        K = rt.matrix(n1, n2)
        ld = K.shape[1]
        bad1 = _nan_rows(X1)
        bad2 = bad1 if X2 is X1 else _nan_rows(X2)
        if not want_dtheta:
            _lib.check(lib.gpx_gram(rt.h, kind, ptr(dX1), n1, dX1.shape[1], ptr(dX2), n2, dX2.shape[1],
                                    dim, sigma, beta_p, 0.0, ptr(K), ld, _lib.GPX_FULL), "gpx_gram")
            return _restore_nan(rt.download(K[:n1, :n2]), bad1, bad2), None, None
        nplanes = len(beta_arr)
        dsig = rt.matrix(n1, n2)
        dbeta = rt.torch.empty((nplanes, max(n1, 1), ld), dtype=rt.torch.float64, device=rt.dev)
        _lib.check(lib.gpx_gram_dtheta(rt.h, kind, ptr(dX1), n1, dX1.shape[1], ptr(dX2), n2,
                                       dX2.shape[1], dim, sigma, beta_p, ptr(K), ptr(dsig), ptr(dbeta),
                                       ld, max(n1, 1) * ld), "gpx_gram_dtheta")
        return (_restore_nan(rt.download(K[:n1, :n2]), bad1, bad2), _restore_nan(rt.download(dsig[:n1, :n2]), bad1, bad2),
                _restore_nan(rt.download(dbeta[:, :n1, :n2]), bad1, bad2))

    def gram_only(self, X1, X2=None):
        """K(X1, X2) as a host array without the derivative planes ``__call__`` also builds and downloads
        (kernels.py:38-39, 80-81): what posterior()/predict() need."""
        X1 = _as_2d(X1, "X1")
        X2 = X1 if X2 is None else _as_2d(X2, "X2")
        assert X1.shape[1] == X2.shape[1], "X1 and X2 must have the same number of features"
        return self._evaluate(X1, X2, want_dtheta=False)[0]

    @on_device
    def device_gram(self, X, diag_add=0.0):
        """K(X, X) + diag_add * I as a device-resident matrix (full, symmetric): what the estimators use
        when the host copy that ``__call__`` returns is not needed.  Same argument checks as
        ``__call__``; no derivative planes, nothing crosses to the host."""
        X = _as_2d(X, "X1")
        rt = runtime()
        rt.bind_stream()
        n, dim = X.shape
        kind, sigma, beta = self.gram_params(dim)
        _, beta_p = _keep = _lib.host_doubles(beta)
        dX = rt.upload_matrix(X)
        K = rt.matrix(n, n)
        _lib.check(rt.lib.gpx_gram(rt.h, kind, ptr(dX), n, dX.shape[1], ptr(dX), n, dX.shape[1], dim, sigma, beta_p,
                                   float(diag_add), ptr(K), K.shape[1], _lib.GPX_FULL), "gpx_gram")
        return K

    def _call(self, X1, X2, diag, dim_check=None):
        X1 = _as_2d(X1, "X1")
        X2 = X1 if X2 is None else _as_2d(X2, "X2")
        n1, d1 = X1.shape
        n2, d2 = X2.shape
        assert d1 == d2, "X1 and X2 must have the same number of features"
        if dim_check is not None:
            assert d1 == dim_check, "input dimension does not match len(beta)"
        if diag:
            m = min(n1, n2)
            nb = len(self._beta_vector(d1))
            dbeta = np.zeros(m) if self.kind == "rbf" else np.zeros((nb, m))
            return KernelValue(float(self.sigma) * np.ones(m), {"sigma": np.ones(m), "beta": dbeta})
        # K now; the derivative planes when (and if) somebody reads dtheta.  The parameters are frozen at call
        # time: a later set_params must not change what this KernelValue's dtheta means.
        frozen = clone_params(self)
        K = frozen._evaluate(X1, X2, want_dtheta=False)[0]

        def build():
            _, dsig, dbeta = frozen._evaluate(X1, X2)
            return {"sigma": dsig, "beta": dbeta[0] if frozen.kind == "rbf" else dbeta}

        return KernelValue(K, LazyDtheta(build))


class RBF(_StationaryKernel):
    """k(x, x') = sigma * exp(-0.5 * beta * |x - x'|^2)   (kernels.py:9-46)."""

    kind = "rbf"
    param_bounds = {"beta": (_LOWER_BOUND, None), "sigma": (_LOWER_BOUND, None)}

    def __init__(self, sigma=1, beta=1):
        self.sigma = sigma
        self.beta = beta

    def _beta_vector(self, dim):
        return np.array([float(self.beta)])

    def __call__(self, X1, X2=None, diag=False):
        return self._call(X1, X2, diag)


class ARD(_StationaryKernel):
    """k(x, x') = sigma * exp(-0.5 * sum_k beta_k (x_k - x'_k)^2)   (kernels.py:49-88)."""

    kind = "ard"

    def __init__(self, sigma=1, beta=np.ones(3)):
        self.sigma = sigma
        self.beta = np.array(beta)
        self.dim = len(beta)
        self.param_bounds = {"sigma": (_LOWER_BOUND, None),
                             "beta": [(_LOWER_BOUND, None)] * len(beta)}

    def _sync_beta(self):
        # set_params(cov__beta=ndarray) replaces the attribute (kernels.py:62-64)
        self.beta = np.array(self.beta)
        assert self.beta.ndim == 1
        self.dim = len(self.beta)

    def _beta_vector(self, dim):
        self._sync_beta()
        assert dim == self.dim, "input dimension does not match len(beta)"
        return np.asarray(self.beta, dtype=np.float64)

    def __call__(self, X1, X2=None, diag=False):
        self._sync_beta()
        return self._call(X1, X2, diag, dim_check=self.dim)
