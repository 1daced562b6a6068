"""GPU parity of the EP classifier (site loop, Cholesky and solves all on the device) against fixtures made by
the real reference with _ep(K, shuffle=False) (expectation_propagation.py:25-108)."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["g4_ep_moons_n50", "g4_ep_moons_n200", "g4_ep_moons_n512", "g4_ep_moons_n1024"])
def test_ep_matches_reference(name):
    """n = 512 and 1024 cross the 256-site block of the delayed rank-1 updates (gpx_ep_sweep) against the
    REFERENCE itself, not only the oracle."""
    from gaussian_process_amd import GaussianProcessExpectationPropagation, RBF
    g = load_golden(name)
    ep = GaussianProcessExpectationPropagation(RBF())
    ep.Xtr, ep.ytr = g["Xtr"], g["ytr"]
    ep.n, ep.dim = g["Xtr"].shape
    K = ep.cov(ep.Xtr).K
    if g["K"].size:
        assert relerr(K, g["K"]) < 1e-13
        K = g["K"]
    else:                 # large cases hold sampled entries of the reference's K only
        idx = g["K_idx"]
        assert relerr(K[idx[:, 0], idx[:, 1]], g["K_samples"]) < 1e-13
    ep._ep(K, shuffle=False)
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


@pytest.mark.parametrize("kind", ["rbf", "ard"])
def test_ep_gradient_rbf_and_ard(kind):
    """Device EP gradient (U = L^-1 S by TRSM, U^T U on the matrix cores, the two trace terms by the fused
    trace kernel) against the oracle's restatement of expectation_propagation.py:161-166, plane by plane, from
    the SAME fitted state, with splitted_terms True and False.  ARD: the per-dimension gradient (deliberate
    deviation, see the method's docstring) -- parity pinned by the oracle only."""
    from gaussian_process_amd import GaussianProcessExpectationPropagation, RBF, ARD
    from oracle import gp_oracle as O
    g = load_golden("g4_ep_moons_n200")
    X, y = g["Xtr"], g["ytr"]
    n, d = X.shape
    beta = np.array([0.7, 1.9]) if kind == "ard" else np.array([1.3])
    cov = ARD(sigma=1.4, beta=beta) if kind == "ard" else RBF(sigma=1.4, beta=float(beta[0]))
    ep = GaussianProcessExpectationPropagation(cov)
    ep.Xtr, ep.ytr, ep.n, ep.dim = X, y, n, d
    K, dsig, dbeta = O.cov(kind, X, None, 1.4, beta, want_grad=True)
    ep._ep(K.copy(), shuffle=False)
    state = dict(L=ep.L, S=ep.S.ravel(), K=K + 1e-8 * np.eye(n), nu=ep.nu_tilde)
    want_sigma = O.ep_grad(state, dsig)
    want_beta = [O.ep_grad(state, dbeta)] if kind == "rbf" else [O.ep_grad(state, dbeta[k]) for k in range(d)]
    split = ep.grad_log_marginal_likelihood(splitted_terms=True)
    assert list(split) == ["cov__sigma", "cov__beta"]
    tol = lambda v: 1e-8 * max(1.0, abs(v))
    for i in range(2):
        assert abs(split["cov__sigma"][i] - want_sigma[i]) < tol(want_sigma[i])
    if kind == "rbf":
        for i in range(2):
            assert abs(split["cov__beta"][i] - want_beta[0][i]) < tol(want_beta[0][i])
    else:
        for i in range(2):
            assert np.shape(split["cov__beta"][i]) == (d,)
            for k in range(d):
                assert abs(split["cov__beta"][i][k] - want_beta[k][i]) < tol(want_beta[k][i])
    total = ep.grad_log_marginal_likelihood()
    assert abs(total["cov__sigma"] - sum(want_sigma)) < tol(sum(want_sigma))
    if kind == "rbf":
        assert np.ndim(total["cov__beta"]) == 0
        assert abs(total["cov__beta"] - sum(want_beta[0])) < tol(sum(want_beta[0]))
    else:
        assert np.shape(total["cov__beta"]) == (d,)      # what base.py:38-39 flattens into the optimiser's vector
        for k in range(d):
            assert abs(total["cov__beta"][k] - sum(want_beta[k])) < tol(sum(want_beta[k]))


def test_ep_notebook_fit_score():
    """examples/expectation_propagation.ipynb: fit() with the RNG-driven site order scores 0.92."""
    from gaussian_process_amd import GaussianProcessExpectationPropagation, RBF
    g = load_golden("g4_ep_moons_n50")
    np.random.seed(0)
    ep = GaussianProcessExpectationPropagation(RBF())
    ep.fit(g["Xtr"], g["ytr"])
    assert abs(ep.score(g["Xte"], g["yte"]) - 0.92) < 1e-12
    assert abs(ep.log_marginal_likelihood() - float(g["lml"])) < 1e-4
    # fit() keeps K and Sigma on the device; the reference's host attributes appear on first use
    n = len(g["ytr"])
    assert ep._K_host is None and ep._Sigma_host is None
    assert relerr(ep.K, g["K"] + 1e-8 * np.eye(n)) < 1e-13
    assert ep.Sigma.shape == (n, n) and np.array_equal(np.diag(ep.Sigma), ep._diag_Sigma())
    assert relerr(np.diag(ep.L), g["diagL"]) < 1e-4


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
    # a site visited again -- right away and a few positions later -- must not share a launch group with its
    # first visit (the kernel takes 8 pairwise distinct sites per launch; the host splits the group)
    order = np.concatenate([order[:5], order[4:5], order[5:9], order[2:3], order[9:]])
    tau, nu, Sg, m = tau0.copy(), nu0.copy(), Sigma.copy(), mu.copy()
    for i in order:
        O.ep_site_update(int(i), y[i], Sg, m, tau, nu)
    rt = runtime()
    rt.bind_stream()
    dS, dy, dt, dn, work = rt.upload_matrix(Sigma), rt.upload_vector(y), rt.upload_vector(tau0), \
        rt.upload_vector(nu0), rt.vector(int(rt.lib.gpx_ep_sweep_work_elems(n)))
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


@pytest.mark.parametrize("name", ["g5_distill_rho100", "g5_distill_rho050", "g5_distill_rho000"])
def test_distillation_matches_reference(name):
    """distillation.py through the device EP path (2n x 2n Kronecker prior, n fixed soft-label sites)
    against fixtures made by the reference: deterministic order via _ep(shuffle=False), then the public
    fit() under the same NumPy seed (examples/distillation_illustration.ipynb)."""
    from gaussian_process_amd import GaussianProcessDistillation, RBF
    g = load_golden(name)
    rho, sigma = float(g["rho"]), float(g["sigma"])
    gpd = GaussianProcessDistillation(cov=RBF(1, 1), sigma=sigma, rho=rho)
    gpd.Xtr, gpd.ytr, gpd.str = g["Xtr"], g["ytr"], g["s_tr"]
    gpd.n, gpd.dim = g["Xtr"].shape
    n = gpd.n
    K_ = gpd.cov(gpd.Xtr).K
    assert relerr(K_, g["K_"]) < 1e-13
    gpd._ep(np.kron(np.array([[1, rho], [rho, 1]]), g["K_"]), fix_index=np.arange(n, 2 * n),
            nu_fixed=g["s_tr"] / sigma, tau_fixed=np.ones(n) / sigma, shuffle=False)
    assert np.array_equal(gpd.tau_tilde[n:], np.ones(n) / sigma)      # fixed sites untouched
    assert relerr(gpd.tau_tilde, g["tau"]) < 1e-7
    assert relerr(gpd.nu_tilde, g["nu"]) < 1e-7
    assert relerr(gpd.mu, g["mu"]) < 1e-7
    assert relerr(np.diag(gpd.Sigma), g["diagSigma"]) < 1e-7
    assert relerr(np.diag(gpd.L), g["diagL"]) < 1e-8
    assert relerr(gpd.z, g["z"]) < 1e-7
    pm, pv = gpd.posterior(g["Xgrid"])
    assert relerr(pm, g["grid_mean"]) < 1e-7 and np.max(np.abs(pv - g["grid_var"])) < 1e-7
    assert relerr(gpd.decision_function(g["Xte"]), g["decision"]) < 1e-7
    assert abs(gpd.score(g["Xte"], g["yte"]) - float(g["score"])) < 1e-12
    lml = gpd.log_marginal_likelihood()
    assert abs(lml - float(g["lml"])) < 1e-7 * abs(float(g["lml"]))
    assert len(gpd.log_marginal_likelihood(splitted_terms=True)) == 6
    # the gradient (the reference raises here): both trace terms against NumPy from the fitted state
    grads = gpd.grad_log_marginal_likelihood(splitted_terms=True)
    assert set(grads) == {"rho", "sigma", "cov__sigma", "cov__beta"}
    K = np.kron(np.array([[1, rho], [rho, 1]]), K_) + 1e-8 * np.eye(2 * n)
    S = np.sqrt(gpd.tau_tilde)
    U = np.linalg.solve(np.linalg.cholesky(np.eye(2 * n) + S[:, None] * K * S[None, :]), np.diag(S))
    b = gpd.nu_tilde - U.T @ U @ K @ gpd.nu_tilde
    dK = np.kron(np.array([[0.0, 1.0], [1.0, 0.0]]), K_)
    assert abs(grads["rho"][0] - 0.5 * b @ dK @ b) < 1e-7 * max(1.0, abs(0.5 * b @ dK @ b))
    assert abs(grads["rho"][1] - 0.5 * np.trace(U.T @ U @ dK)) < 1e-7 * max(1.0, abs(0.5 * np.trace(U.T @ U @ dK)))
    # the other parameters: the Kronecker planes of distillation.py:70-79 built on the host (the device path never forms them)
    from oracle import gp_oracle as O
    _, dsig, dbeta = O.rbf_cov(g["Xtr"], None, 1.0, 1.0, want_grad=True)
    Kt = np.array([[1, rho], [rho, 1]])
    planes = {"sigma": np.kron(np.array([[0.0, 0.0], [0.0, 1.0]]), np.eye(n)), "cov__sigma": np.kron(Kt, dsig),
              "cov__beta": np.kron(Kt, dbeta)}
    UtU = U.T @ U
    for name, v in planes.items():
        want = [0.5 * b @ v @ b, 0.5 * np.sum(UtU * v.T)]
        for i in range(2):
            assert abs(grads[name][i] - want[i]) < 1e-7 * max(1.0, abs(want[i])), (name, i, grads[name][i], want[i])
    total = gpd.grad_log_marginal_likelihood()
    assert abs(total["cov__beta"] - sum(grads["cov__beta"])) < 1e-12 * max(1.0, abs(total["cov__beta"]))
    # public API, RNG-driven site order
    np.random.seed(0)
    pub = GaussianProcessDistillation(cov=RBF(1, 1), sigma=sigma, rho=rho)
    pub.fit(g["Xtr"], g["ytr"], g["s_tr"])
    qm, qv = pub.posterior(g["Xgrid"])
    assert relerr(pub.tau_tilde, g["pub_tau"]) < 1e-6 and relerr(qm, g["pub_grid_mean"]) < 1e-6
    assert abs(pub.score(g["Xte"], g["yte"]) - float(g["pub_score"])) < 1e-12
    pub.set_params(rho=0.25)
    pub.refit()
    assert pub.tau_tilde.shape == (2 * n,)
