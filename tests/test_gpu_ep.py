"""GPU parity of the EP classifier (site loop, Cholesky and solves all on the device) against fixtures made by
the real reference with _ep(K, shuffle=False) (expectation_propagation.py:25-108)."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["g4_ep_moons_n50", "g4_ep_moons_n200"])
def test_ep_matches_reference(name):
    from gaussian_process_amd import GaussianProcessExpectationPropagation, RBF
    g = load_golden(name)
    ep = GaussianProcessExpectationPropagation(RBF())
    ep.Xtr, ep.ytr = g["Xtr"], g["ytr"]
    ep.n, ep.dim = g["Xtr"].shape
    K = ep.cov(ep.Xtr).K
    assert relerr(K, g["K"]) < 1e-13
    ep._ep(g["K"], shuffle=False)
    assert relerr(ep.tau_tilde, g["tau"]) < 1e-8
    assert relerr(ep.nu_tilde, g["nu"]) < 1e-8
    assert relerr(ep.mu, g["mu"]) < 1e-8
    assert relerr(np.diag(ep.Sigma), g["diagSigma"]) < 1e-8
    assert relerr(np.diag(ep.L), g["diagL"]) < 1e-8
    assert relerr(ep.z, g["z"]) < 1e-8
    lml = ep.log_marginal_likelihood()
    assert abs(lml - float(g["lml"])) < 1e-8 * abs(float(g["lml"]))
    assert len(ep.log_marginal_likelihood(splitted_terms=True)) == 7
    pm, pv = ep.posterior(g["Xte"])
    assert relerr(pm, g["post_mean"]) < 1e-8 and relerr(pv, g["post_var"]) < 1e-8
    assert relerr(ep.decision_function(g["Xte"]), g["decision"]) < 1e-8
    assert abs(ep.score(g["Xte"], g["yte"]) - float(g["score"])) < 1e-12
    grads = ep.grad_log_marginal_likelihood()
    assert abs(grads["cov__sigma"] - float(g["grad_cov_sigma"])) < 1e-6 * abs(float(g["grad_cov_sigma"]))
    assert abs(grads["cov__beta"] - float(g["grad_cov_beta"])) < 1e-6 * abs(float(g["grad_cov_beta"]))


def test_ep_notebook_fit_score():
    """examples/expectation_propagation.ipynb: fit() with the RNG-driven site order scores 0.92."""
    from gaussian_process_amd import GaussianProcessExpectationPropagation, RBF
    g = load_golden("g4_ep_moons_n50")
    np.random.seed(0)
    ep = GaussianProcessExpectationPropagation(RBF())
    ep.fit(g["Xtr"], g["ytr"])
    assert abs(ep.score(g["Xte"], g["yte"]) - 0.92) < 1e-12
    assert abs(ep.log_marginal_likelihood() - float(g["lml"])) < 1e-4


@pytest.mark.parametrize("n", [37, 200, 515])
def test_ep_sweep_matches_oracle_site_by_site(n):
    """gpx_ep_sweep (site loop on the device) against the oracle's sequential
    ep_site_update over one shuffled sweep with some sites skipped: tau, nu and Sigma afterwards."""
    import ctypes as C
    from gaussian_process_amd import _lib
    from gaussian_process_amd.device import ptr, runtime
    from oracle import gp_oracle as O
    rng = np.random.default_rng(n)
    X = rng.uniform(-2, 2, (n, 2))
    y = np.sign(X[:, 0] * X[:, 1] + 0.1 * rng.standard_normal(n))
    K = O.cov("rbf", X, None, 1.0, 1.0)[0] + 1e-8 * np.eye(n)
    tau0 = rng.uniform(0.0, 0.5, n)
    nu0 = rng.standard_normal(n) * 0.1
    _, Sigma, mu, _ = O.ep_recompute(K, tau0, nu0)
    order = rng.permutation(n)[: max(1, (3 * n) // 4)]
    tau, nu, Sg, m = tau0.copy(), nu0.copy(), Sigma.copy(), mu.copy()
    for i in order:
        O.ep_site_update(int(i), y[i], Sg, m, tau, nu)
    rt = runtime()
    rt.bind_stream()
    dS, dy, dt, dn, work = rt.upload_matrix(Sigma), rt.upload_vector(y), rt.upload_vector(tau0), \
        rt.upload_vector(nu0), rt.vector(n + 4)
    o32 = np.ascontiguousarray(order, dtype=np.int32)
    _lib.check(rt.lib.gpx_ep_sweep(rt.h, n, ptr(dS), dS.shape[1], ptr(dy), ptr(dt), ptr(dn),
                                   o32.ctypes.data_as(C.POINTER(C.c_int)), len(o32), ptr(work)), "gpx_ep_sweep")
    assert relerr(rt.download(dt[:n]), tau) < 1e-10
    assert relerr(rt.download(dn[:n]), nu) < 1e-10
    assert relerr(np.tril(rt.download(dS[:n, :n])), np.tril(Sg)) < 1e-10     # lower triangle is the contract
    # bad site index is rejected before anything is launched for it
    bad = np.array([n], dtype=np.int32)
    assert rt.lib.gpx_ep_sweep(rt.h, n, ptr(dS), dS.shape[1], ptr(dy), ptr(dt), ptr(dn),
                               bad.ctypes.data_as(C.POINTER(C.c_int)), 1, ptr(work)) == -1
