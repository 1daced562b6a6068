"""Pins oracle/gp_oracle.py against outputs of the real reference (tests/golden/*.npz,
made by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import gp_oracle as O

REG_CASES = ["g1_rbf1d_n15", "g1_rbf1d_n200", "g2_ard2d", "g3_ard8_n256", "g3_ard8_n1024"]
TOL = 1e-9  # north_star tolerance (relative, fp64)


def theta(g):
    return str(g["kind"]), float(g["cov_sigma"]), g["cov_beta"], float(g["noise"])


@pytest.mark.parametrize("name", REG_CASES)
def test_cov_matches_reference(name):
    g = load_golden(name)
    kind, s, b, _ = theta(g)
    if "K" not in g:
        K, _, _ = O.cov(kind, g["X"], None, s, b)
        idx = g["K_idx"]
        assert relerr(K[idx[:, 0], idx[:, 1]], g["K_samples"]) < 1e-13
        assert abs(K.sum() - g["K_sum"]) / abs(g["K_sum"]) < 1e-12
        return
    K, dsig, dbeta = O.cov(kind, g["X"], None, s, b, want_grad=True)
    assert relerr(K, g["K"]) < 1e-13
    assert relerr(dsig, g["dsigma"]) < 1e-13
    assert dbeta.shape == g["dbeta"].shape
    assert relerr(dbeta, g["dbeta"]) < 1e-12
    Ks, _, _ = O.cov(kind, g["X"], g["Xs"], s, b)
    assert relerr(Ks, g["Ks"]) < 1e-13


@pytest.mark.parametrize("name", REG_CASES + ["g3_ard8_n4096"])
def test_lml_alpha_predict_match_reference(name):
    g = load_golden(name)
    kind, s, b, noise = theta(g)
    lml, a, L = O.gpr_lml(kind, g["X"], g["y"], s, b, noise)
    assert abs(lml - float(g["lml"])) <= TOL * abs(float(g["lml"]))
    assert relerr(np.diagonal(L), g["diagL"]) < TOL
    assert relerr(a, g["a"]) < TOL
    assert relerr(O.gpr_alpha(L, g["y"]), g["alpha"]) < TOL
    mean, pcov = O.gpr_predict(kind, g["X"], g["y"], g["Xs"], s, b, noise, L=L)
    assert relerr(mean, g["mean"]) < TOL
    assert relerr(pcov, g["pcov"]) < TOL
    _, dvar = O.gpr_predict(kind, g["X"], g["y"], g["Xs"], s, b, noise, L=L, full_cov=False)
    assert relerr(dvar, np.diagonal(g["pcov"])) < 1e-8


@pytest.mark.parametrize("name", REG_CASES)
@pytest.mark.parametrize("spd", [False, True])
def test_grad_matches_reference(name, spd):
    g = load_golden(name)
    kind, s, b, noise = theta(g)
    grads, alpha = O.gpr_grad(kind, g["X"], g["y"], s, b, noise, spd_inverse=spd)
    assert list(grads) == ["sigma", "cov__sigma", "cov__beta"]
    # traces of A against dK: cancellation makes them relative to the gradient scale
    scale = max(np.abs(g["grad_sigma"]).max(), np.abs(g["grad_cov_sigma"]).max(),
                np.abs(g["grad_cov_beta"]).max())
    tol = 1e-8
    assert np.abs(grads["sigma"] - g["grad_sigma"]).max() <= tol * scale
    assert np.abs(grads["cov__sigma"] - g["grad_cov_sigma"]).max() <= tol * scale
    assert grads["cov__beta"].shape == g["grad_cov_beta"].shape
    assert np.abs(grads["cov__beta"] - g["grad_cov_beta"]).max() <= tol * scale
    assert relerr(alpha, g["alpha"]) < TOL


def test_known_answers_from_survey():
    """Values quoted in SURVEY.md section 4 / 8c (reproduced notebook outputs)."""
    g = load_golden("g1_rbf1d_n15")
    assert abs(float(g["lml"]) - (-11.80896075441085)) < 1e-10
    g = load_golden("g2_ard2d")
    assert abs(float(g["lml"]) - (-143.37940612555946)) < 1e-9
    # examples/ARD_regression_2d.ipynb lines 77 / 111 / 142 (stored notebook outputs)
    assert abs(float(g["rmse_before"]) - 0.285960216844) < 1e-6
    assert abs(float(g["rmse_after"]) - 0.208740245085) < 1e-4
    assert np.allclose(g["eb_beta"], [1.14651806, 3.7085029], rtol=1e-4)


@pytest.mark.parametrize("name", ["g4_ep_moons_n50", "g4_ep_moons_n200", "g4_ep_moons_n512"])
def test_ep_matches_reference(name):
    g = load_golden(name)
    K = g["K"]
    if K.size == 0:       # large cases hold sampled entries of K only
        K = O.rbf_cov(g["Xtr"], None, 1.0, 1.0, want_grad=False)[0]
        idx = g["K_idx"]
        assert relerr(K[idx[:, 0], idx[:, 1]], g["K_samples"]) < 1e-13
        assert abs(K.sum() - float(g["K_sum"])) < 1e-12 * abs(float(g["K_sum"]))
    st = O.ep_fit(K, g["ytr"])
    assert relerr(st["tau"], g["tau"]) < 1e-8
    assert relerr(st["nu"], g["nu"]) < 1e-8
    assert relerr(st["mu"], g["mu"]) < 1e-8
    assert relerr(np.diag(st["Sigma"]), g["diagSigma"]) < 1e-8
    assert relerr(np.diag(st["L"]), g["diagL"]) < 1e-8
    assert relerr(st["z"], g["z"]) < 1e-8
    assert abs(O.ep_lml(st, g["ytr"]) - float(g["lml"])) < 1e-8 * abs(float(g["lml"]))
    Ks, _, _ = O.rbf_cov(g["Xtr"], g["Xte"], 1.0, 1.0, want_grad=False)
    pm, pv = O.ep_posterior(st, Ks, np.ones(g["Xte"].shape[0]))
    assert relerr(pm, g["post_mean"]) < 1e-8
    assert relerr(pv, g["post_var"]) < 1e-8


@pytest.mark.parametrize("name", ["g5_distill_rho100", "g5_distill_rho050", "g5_distill_rho000"])
def test_distillation_matches_reference(name):
    """distillation.py:22-63 (EP with the soft-label sites fixed) against the reference's own outputs."""
    g = load_golden(name)
    rho, sigma = float(g["rho"]), float(g["sigma"])
    st = O.distillation_fit(g["K_"], g["ytr"], g["s_tr"], rho, sigma)
    n = len(g["ytr"])
    assert np.array_equal(st["tau"][n:], np.ones(n) / sigma)
    for key, ref in (("tau", "tau"), ("nu", "nu"), ("mu", "mu"), ("z", "z")):
        assert relerr(st[key], g[ref]) < 1e-7, key
    assert relerr(np.diag(st["Sigma"]), g["diagSigma"]) < 1e-7
    assert relerr(np.diag(st["L"]), g["diagL"]) < 1e-8
    Ks_, _, _ = O.rbf_cov(g["Xtr"], g["Xgrid"], 1.0, 1.0, want_grad=False)
    pm, pv = O.distillation_posterior(st, Ks_, np.ones(g["Xgrid"].shape[0]), rho)
    assert relerr(pm, g["grid_mean"]) < 1e-7
    assert np.max(np.abs(pv - g["grid_var"])) < 1e-7
    assert abs(O.distillation_lml(st, g["ytr"]) - float(g["lml"])) < 1e-7 * abs(float(g["lml"]))
    assert str(g["grad_raises"]) == "ValueError"       # the reference's gradient cannot be evaluated
