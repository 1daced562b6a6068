"""GPU parity of the EP classifier (host site loop + GPU Cholesky/solves) against fixtures made by
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
