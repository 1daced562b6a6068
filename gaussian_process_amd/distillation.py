"""GP distillation by expectation propagation (reference: gaussian_process/distillation.py).

A two-task model over the same inputs: task 0 carries the hard labels y through the probit EP
sites, task 1 the teacher's soft labels s as FIXED Gaussian sites (nu = s / sigma, tau = 1 / sigma);
the tasks are tied by the 2 x 2 task kernel [[1, rho], [rho, 1]], so the joint prior is the
2n x 2n Kronecker matrix (distillation.py:27-35).  All linear algebra and the site sweep run on the
MI355X through the inherited EP machinery (expectation_propagation.py of this package):

  fit                     distillation.py:22-35   -> _ep(kron(Kt, K_), fixed sites n..2n-1)
  posterior               distillation.py:40-44   -> gpx_trsm on the 2n x m stacked cross-covariance
  log_marginal_likelihood distillation.py:46-63   formula kept as written (trace(L) in t0 and the
                                                  soft-label half of the cavity terms in t3 look
                                                  unintended; not "fixed" here)
  grad_log_marginal_...   distillation.py:65-81   the reference raises ValueError at
                                                  expectation_propagation.py:163 (adds an n x n jitter
                                                  to the 2n x 2n K); here the jitter has the size of K
                                                  and the two trace terms per parameter are returned.
"""
import numpy as np
from scipy.special import log_ndtr
from sklearn.utils import check_X_y

from .expectation_propagation import GaussianProcessExpectationPropagation


class GaussianProcessDistillation(GaussianProcessExpectationPropagation):
    """Gaussian process distillation with the EP algorithm; drop-in for the reference."""

    def __init__(self, cov, rho, sigma):
        self.cov = cov
        self.rho = rho
        self.sigma = sigma
        self.param_bounds = {"rho": (0, 1), "sigma": (1e-5, None)}
        for name, bound in self.cov.param_bounds.items():
            self.param_bounds["cov__" + name] = bound

    def _task_kernel(self):
        return np.array([[1.0, self.rho], [self.rho, 1.0]])

    def fit(self, X, y, s):
        self.Xtr, self.ytr = check_X_y(X, y)
        _, self.str = check_X_y(X, s, y_numeric=True)
        self.n, self.dim = self.Xtr.shape
        K = np.kron(self._task_kernel(), self.cov(self.Xtr).K)
        self._ep(K, fix_index=np.arange(self.n, 2 * self.n), nu_fixed=self.str / self.sigma,
                 tau_fixed=np.ones(self.n) / self.sigma)

    def refit(self):
        self.fit(self.Xtr, self.ytr, self.str)

    def posterior(self, X):
        Ks_ = self._cross_cov(X)
        Kss = self.cov(X, diag=True).K
        return self._posterior(np.vstack([Ks_, self.rho * Ks_]), Kss)

    def log_marginal_likelihood(self, eps=1e-7, splitted_terms=False):
        n = self.n
        dS = self._diag_Sigma()
        tau_cav = 1 / dS - self.tau_tilde
        nu_cav = self.mu / dS - self.nu_tilde
        t0 = -np.log(np.trace(self.L))
        t1 = np.sum(np.log(self.S))
        # (L diag(S))^-1 nu = diag(S)^-1 L^-1 nu: one triangular solve on the device
        w = self._solve_vec(self._dv, self.nu_tilde, trans=0) / self.S.ravel()
        t2 = -0.5 * w.dot(w)
        t3 = np.sum(log_ndtr(self.ytr * nu_cav[n:] / np.sqrt(tau_cav[n:] * (1 + tau_cav[n:]))))
        assert np.all(tau_cav[n:] > eps)
        assert np.all(self.tau_tilde[n:] > eps)
        t4 = 0.5 * np.sum(np.log((1 / tau_cav + 1 / self.tau_tilde)[n:]))
        t5 = 0.5 * np.sum(((nu_cav / tau_cav + self.nu_tilde / self.tau_tilde) ** 2
                           * (1 / tau_cav + 1 / self.tau_tilde))[n:])
        terms = (t0, t1, t2, t3, t4, t5)
        return terms if splitted_terms else sum(terms)

    def grad_log_marginal_likelihood(self, splitted_terms=False):
        n = self.n
        kernel = self.cov(self.Xtr)
        Kt = self._task_kernel()
        K = np.kron(Kt, kernel.K)
        zero = np.zeros((n, n))
        K_grads = {"rho": np.kron(np.array([[0.0, 1.0], [1.0, 0.0]]), kernel.K),
                   "sigma": np.block([[zero, zero], [zero, np.eye(n)]])}
        for name, v in kernel.dtheta.items():
            K_grads["cov__" + name] = np.kron(Kt, v)
        return {k: v if splitted_terms else sum(v)
                for k, v in self._grad_log_marginal_likelihood(K, K_grads).items()}
