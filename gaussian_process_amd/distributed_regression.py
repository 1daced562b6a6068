"""The reference estimator's interface (regression.py:8-76) over the ranks of torch.distributed.

``DistributedGaussianProcessRegression(cov, sigma)`` has the constructor, methods and return values of
``GaussianProcessRegression``; underneath every rank of the default process group (one process per
GPU, backend "nccl" = RCCL) runs ``distributed.DistributedGP``.  Usage contract: every rank builds
the estimator with the same data and hyper-parameters and calls the same methods in the same
order (they contain collectives); every rank gets the same return values -- LML and alpha are
computed redundantly from identical inputs, the gradient is all-reduced -- so L-BFGS-B inside
``empirical_bayes`` takes identical steps everywhere and the ranks stay in lockstep.
"""
import numpy as np

from .distributed import DistributedGP, GpuOps, default_block
from .regression import GaussianProcessRegression


class DistributedGaussianProcessRegression(GaussianProcessRegression):
    #: keyword options of DistributedGP (block, grid, update_k, grad_block); None = its defaults for the world size
    dist_options = None
    #: callable returning the per-rank primitive set; tests substitute a NumPy double
    ops_factory = None

    def _engine(self):
        gp = getattr(self, "_gp", None)
        if gp is None:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("DistributedGaussianProcessRegression needs torch.distributed.init_process_group "
                                   "(one process per GPU, backend 'nccl')")
            rank, world = dist.get_rank(), dist.get_world_size()
            if self.ops_factory is not None:
                ops = type(self).ops_factory()
            else:
                from .device import runtime
                rt = runtime()
                rt.bind_stream()
                ops = GpuOps(rt)
            opts = dict(self.dist_options or {})
            opts.setdefault("block", default_block(world))
            gp = self._gp = DistributedGP(ops, dist, rank, world, **opts)
        return gp

    def fit(self, X, y, empirical_bayes=False):
        super().fit(X, y, empirical_bayes=False)
        self._uploaded = False
        if empirical_bayes:
            self.empirical_bayes()

    def refit(self):
        pass   # nothing depends on theta at fit time (regression.py:25-26); evaluations are keyed on theta

    def _evaluate(self, need_grad):
        key = self._theta_key()
        st = getattr(self, "_state", None)
        if st is not None and st["key"] == key and (st["grad"] is not None or not need_grad):
            return st
        gp = self._engine()
        if not getattr(self, "_uploaded", False):
            gp.allocate(self.Xtr, self.ytr, key[0], key[1], np.array(key[2]), key[3])
            self._uploaded = True
        if st is None or st["key"] != key:
            st = self._state = None      # the resident factor is overwritten from here on, also if potrf raises
            gp.set_theta(key[0], key[1], np.array(key[2]), key[3])
            gp.build_gram()
            gp.potrf()
            lml = gp.solve_lml()          # raises LinAlgError on every rank if Ky is not positive definite
            st = self._state = dict(key=key, lml=float(lml), grad=None, has_factor=True)
        if need_grad and st["grad"] is None:
            st["grad"] = gp.grad()
        return st

    def alpha_(self):
        self._evaluate(need_grad=False)
        a = self._gp.alpha
        return a.detach().cpu().numpy()[: self.n].copy()

    def predict_with_variance(self, X, full_cov=True, batch=16384):
        """(mean, cov) as regression.py:31-44; ``full_cov=False`` returns the diagonal variance, never
        forms the m x m matrix and streams the test points ``batch`` at a time, as the single-GPU
        estimator does."""
        X = np.ascontiguousarray(np.asarray(X), dtype=np.float64)
        assert X.ndim == 2 and X.shape[1] == self.dim
        self._evaluate(need_grad=False)
        if full_cov:
            return self._gp.predict(X)
        parts = [self._gp.predict(X[i:i + batch], full_cov=False) for i in range(0, max(X.shape[0], 1), batch)]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
