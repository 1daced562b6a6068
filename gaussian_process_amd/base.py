"""Hyper-parameter optimisation driver (reference: gaussian_process/base.py).

Host Python, as in the reference: L-BFGS-B on the negative log marginal likelihood over the
flattened sklearn parameters.  The estimator-facing contract is unchanged (``refit`` /
``log_marginal_likelihood`` / ``grad_log_marginal_likelihood``; parameter names, order and
bounds come from ``get_params`` / ``param_bounds`` as at base.py:22-46).  The GPU estimators
cache one fused (LML, gradient) evaluation per parameter vector, so the optimiser's separate
calls (base.py:35-37) cost one Gram + one factorisation instead of two Grams and three
factorisations.

Fix over the reference: ``random=True`` draws the start point with the right length
(base.py:29 references an undefined name).
"""
from abc import ABCMeta, abstractmethod

import numpy as np
from scipy.optimize import minimize
from sklearn.base import BaseEstimator


def _is_opt_value(v):
    return np.isscalar(v) or isinstance(v, np.ndarray)


class BayesEstimator(BaseEstimator, metaclass=ABCMeta):

    @abstractmethod
    def refit(self):
        """Re-run whatever ``fit`` computes for the current hyper-parameters."""

    @abstractmethod
    def log_marginal_likelihood(self):
        """Scalar log evidence at the current hyper-parameters."""

    @abstractmethod
    def grad_log_marginal_likelihood(self):
        """Dict name -> d LML / d theta at the current hyper-parameters."""

    # -- flattening of the sklearn parameter dict (base.py:60-84) -----------------------
    def get_opt_params(self, param_names):
        values = self.get_params()
        flat = []
        for name in param_names:
            v = values[name]
            if np.isscalar(v):
                flat.append(v)
            elif isinstance(v, np.ndarray):
                flat.extend(v.ravel())
        return np.array(flat)

    def set_opt_params(self, opt_params, param_names):
        values = self.get_params()
        update, pos = {}, 0
        for name in param_names:
            v = values[name]
            if np.isscalar(v):
                update[name] = opt_params[pos]
                pos += 1
            elif isinstance(v, np.ndarray):
                update[name] = np.asarray(opt_params[pos:pos + v.size]).reshape(v.shape)
                pos += v.size
        self.set_params(**update)

    def _opt_bounds(self, param_names):
        bounds = []
        for name in param_names:
            b = self.param_bounds[name]
            if isinstance(b[0], tuple):      # per-dimension list (ARD beta)
                bounds.extend(b)
            else:
                bounds.append(b)
        return bounds

    def empirical_bayes(self, opt_param_names=None, random=False, verbose=False):
        """Type-II maximum likelihood (base.py:21-58).  Keeps the optimum only if it improves."""
        if opt_param_names is None:
            opt_param_names = [k for k, v in self.get_params().items() if _is_opt_value(v)]
        start = self.get_opt_params(opt_param_names)
        start_cost = -self.log_marginal_likelihood()
        x0 = np.random.rand(len(start)) * 3 if random else start

        def cost_and_grad(theta):
            self.set_opt_params(theta, opt_param_names)
            self.refit()
            g = self.grad_log_marginal_likelihood()
            cost = -self.log_marginal_likelihood()
            return cost, -np.concatenate([np.ravel(g[k]) for k in opt_param_names])

        res = minimize(cost_and_grad, x0, method="L-BFGS-B", jac=True,
                       bounds=self._opt_bounds(opt_param_names), options={"disp": verbose})
        if verbose:
            print(res.message)
        self.set_opt_params(res.x if res.fun < start_cost else start, opt_param_names)
        self.optimizer_result_ = res
