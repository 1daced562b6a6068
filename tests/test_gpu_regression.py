"""GPU parity of the drop-in estimators against the golden fixtures produced by the real
reference (tests/golden/*.npz) and against the oracle.  These read like the reference's own
usage (examples/*.ipynb): same constructors, same method names, same return shapes.

Tolerance (north star): LML, alpha, predictive mean / covariance within 1e-9 relative, fp64."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


def make_gp(g):
    from gaussian_process_amd import GaussianProcessRegression, RBF, ARD
    if str(g["kind"]) == "ard":
        cov = ARD(sigma=float(g["cov_sigma"]), beta=g["cov_beta"])
    else:
        cov = RBF(sigma=float(g["cov_sigma"]), beta=float(g["cov_beta"][0]))
    gp = GaussianProcessRegression(cov=cov, sigma=float(g["noise"]))
    assert gp.fit(g["X"], g["y"]) is None      # regression.py:18-23 returns None
    return gp


@pytest.mark.parametrize("name", ["g1_rbf1d_n15", "g1_rbf1d_n200", "g2_ard2d", "g3_ard8_n256",
                                  "g3_ard8_n1024", "g3_ard8_n4096"])
def test_fixed_theta_matches_reference(name):
    g = load_golden(name)
    gp = make_gp(g)
    lml = gp.log_marginal_likelihood()
    assert isinstance(lml, float)
    assert abs(lml - float(g["lml"])) <= TOL * abs(float(g["lml"]))
    assert relerr(gp.alpha_(), g["alpha"]) < TOL
    mean, cov = gp.predict_with_variance(g["Xs"])
    m = g["Xs"].shape[0]
    assert mean.shape == (m,) and cov.shape == (m, m)
    assert relerr(mean, g["mean"]) < TOL
    assert relerr(cov, g["pcov"]) < TOL
    assert relerr(gp.predict(g["Xs"]), g["mean"]) < TOL
    mean2, var = gp.predict_with_variance(g["Xs"], full_cov=False)
    assert relerr(var, np.diagonal(g["pcov"])) < 1e-8
    grads = gp.grad_log_marginal_likelihood()
    assert list(grads) == ["sigma", "cov__sigma", "cov__beta"]
    assert grads["sigma"].shape == (1,) and grads["cov__sigma"].shape == (1,)
    assert grads["cov__beta"].shape == g["grad_cov_beta"].shape
    scale = max(np.abs(g["grad_sigma"]).max(), np.abs(g["grad_cov_sigma"]).max(),
                np.abs(g["grad_cov_beta"]).max())
    # traces of A = alpha alpha^T - Ky^-1 against dK: cancellation => relative to the largest
    for k, ref in (("sigma", g["grad_sigma"]), ("cov__sigma", g["grad_cov_sigma"]),
                   ("cov__beta", g["grad_cov_beta"])):
        assert np.abs(grads[k] - ref).max() <= 1e-8 * scale, k
    # LML again after the gradient pass (cached) and prediction after it: the gradient pass turned the
    # factor into its inverse, so V = L^-1 k* is a triangular multiply (gpx_gpr_predict_inv), no refactorisation
    assert gp.log_marginal_likelihood() == lml
    assert gp._state["has_factor"] is False
    assert relerr(gp.predict(g["Xs"]), g["mean"]) < TOL
    mean3, cov3 = gp.predict_with_variance(g["Xs"])
    assert gp._state["has_factor"] is False and gp._state["grad"] is not None      # still the gradient's state
    assert relerr(mean3, g["mean"]) < TOL and relerr(cov3, g["pcov"]) < TOL
    assert relerr(gp.predict_with_variance(g["Xs"], full_cov=False)[1], np.diagonal(g["pcov"])) < 1e-8


def test_regression_1d_notebook_flow():
    """examples/regression_1d.ipynb cells 2-4: fit, predict, empirical_bayes, LML sweep via set_params."""
    from gaussian_process_amd import GaussianProcessRegression, RBF
    g = load_golden("g1_rbf1d_n15")
    gp = GaussianProcessRegression(cov=RBF(), sigma=0.1)
    gp.fit(g["X"], g["y"])
    assert abs(gp.log_marginal_likelihood() - (-11.80896075441085)) < 1e-9
    gp.empirical_bayes()
    assert abs(gp.log_marginal_likelihood() - float(g["eb_lml"])) < 1e-5
    assert np.allclose([gp.cov.beta, gp.cov.sigma, gp.sigma],
                       [float(g["eb_beta"]), float(g["eb_cov_sigma"]), float(g["eb_noise"])], rtol=1e-3)
    vals = []
    for b in (0.2, 1.0):
        for s in (0.05, 0.5):
            gp.set_params(cov__beta=b, sigma=s)
            vals.append(gp.log_marginal_likelihood())
            lo, _, _ = O.gpr_lml("rbf", g["X"], g["y"], gp.cov.sigma, b, s)
            assert abs(vals[-1] - lo) <= TOL * abs(lo)
    assert len(set(vals)) == 4
    r2 = gp.score(g["X"], g["y"])        # RegressorMixin
    assert np.isfinite(r2)


def test_ard_2d_notebook_empirical_bayes():
    """examples/ARD_regression_2d.ipynb: RMSE 0.28596 -> 0.20874, beta -> [1.1465, 3.7085]."""
    from gaussian_process_amd import GaussianProcessRegression, ARD
    g = load_golden("g2_ard2d")
    gp = GaussianProcessRegression(cov=ARD(beta=[1, 1]))
    gp.fit(g["X"], g["y"])
    rmse0 = np.sqrt(np.mean((gp.predict(g["Xs"]) - g["yte"]) ** 2))
    assert abs(rmse0 - 0.285960216844) < 1e-8
    gp.fit(g["X"], g["y"], empirical_bayes=True)
    rmse1 = np.sqrt(np.mean((gp.predict(g["Xs"]) - g["yte"]) ** 2))
    assert abs(rmse1 - 0.208740245085) < 1e-4
    assert np.allclose(gp.cov.beta, [1.14651806, 3.7085029], rtol=1e-4)
    assert abs(gp.log_marginal_likelihood() - float(g["eb_lml"])) < 1e-5 * abs(float(g["eb_lml"]))


def test_not_positive_definite_raises_linalgerror():
    from gaussian_process_amd import GaussianProcessRegression, RBF
    X = np.zeros((6, 1))
    gp = GaussianProcessRegression(cov=RBF(), sigma=-2.0)   # K + sigma I indefinite
    gp.fit(X, np.arange(6.0))
    with pytest.raises(np.linalg.LinAlgError):
        gp.log_marginal_likelihood()


def test_bad_inputs_raise_like_reference():
    from gaussian_process_amd import GaussianProcessRegression, RBF
    gp = GaussianProcessRegression(cov=RBF())
    with pytest.raises(ValueError):
        gp.fit(np.zeros((4, 1)), np.array([0.0, np.nan, 1.0, 2.0]))   # check_X_y
    with pytest.raises(ValueError):
        gp.fit(np.zeros(4), np.zeros(4))


@pytest.mark.parametrize("n", [8192, 9001, 16384])      # 16384: BASELINE configs[2] at its exact size; 9001: ragged
# blocks everywhere (recursion splits, the tail of the 1024-block solves, the last leaf)
def test_large_size_properties(n):
    """Beyond the sizes the reference can run: size-independent properties.
    ||Ky alpha - y|| / ||y||, ||L L^T - Ky||_F / ||Ky||_F on sampled rows, LML against the oracle
    restatement, gradient by central finite differences of the GPU LML."""
    from gaussian_process_amd import GaussianProcessRegression, ARD
    X, y, Xs, beta = O.synthetic_ard(n, m=300, d=8, seed=1, noise=0.1)
    gp = GaussianProcessRegression(cov=ARD(sigma=1.0, beta=beta), sigma=0.1)
    gp.fit(X, y)
    lml = gp.log_marginal_likelihood()
    alpha = gp.alpha_()
    K, _, _ = O.ard_cov(X, None, 1.0, beta)
    K[np.diag_indices_from(K)] += 0.1
    assert np.linalg.norm(K @ alpha - y) / np.linalg.norm(y) < 1e-11
    lml_o, _, L = O.gpr_lml("ard", X, y, 1.0, beta, 0.1)
    assert abs(lml - lml_o) <= TOL * abs(lml_o)
    assert relerr(alpha, O.gpr_alpha(L, y)) < TOL
    mean, cov = gp.predict_with_variance(Xs)
    mo, co = O.gpr_predict("ard", X, y, Xs, 1.0, beta, 0.1, L=L)
    assert relerr(mean, mo) < TOL and relerr(cov, co) < TOL
    assert np.allclose(cov, cov.T, rtol=0, atol=0)
    g = gp.grad_log_marginal_likelihood()
    h = 1e-4
    gp.set_params(sigma=0.1 + h)
    up_ = gp.log_marginal_likelihood()
    gp.set_params(sigma=0.1 - h)
    dn = gp.log_marginal_likelihood()
    assert abs((up_ - dn) / (2 * h) - g["sigma"][0]) < 1e-4 * abs(g["sigma"][0])


def test_streaming_predict_matches_one_shot_and_oracle():
    """predict() / full_cov=False tile over the test points: any batch size gives the
    one-shot numbers, and those match the oracle's mean and diagonal variance."""
    from gaussian_process_amd import ARD, GaussianProcessRegression
    from oracle import gp_oracle as O
    X, y, Xs, beta = O.synthetic_ard(600, m=333, d=4, seed=5)
    gp = GaussianProcessRegression(ARD(sigma=1.2, beta=beta), sigma=0.05)
    gp.fit(X, y)
    mean_o, var_o = O.gpr_predict("ard", X, y, Xs, 1.2, beta, 0.05, full_cov=False)
    m1, v1 = gp.predict_with_variance(Xs, full_cov=False)
    m2, v2 = gp.predict_with_variance(Xs, full_cov=False, batch=50)      # 7 ragged batches
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
    assert relerr(m1, mean_o) < 1e-9 and relerr(v1, var_o) < 1e-9
    assert np.array_equal(gp.predict(Xs), m1)
    mean_f, cov_f = gp.predict_with_variance(Xs)
    assert relerr(mean_f, mean_o) < 1e-9 and relerr(np.diag(cov_f), var_o) < 1e-9


def test_generate_gp_same_X_as_reference_and_y_is_L_z():
    """datasets.generate_gp: X must be the reference's draw for the seed (golden G1 holds it); the
    sample is y = L z with L the device Cholesky factor of K + noise I (values unpinned by design)."""
    import scipy.linalg as sla
    from gaussian_process_amd import RBF
    from gaussian_process_amd.datasets import generate_gp
    from oracle import gp_oracle as O
    g = load_golden("g1_rbf1d_n15")
    X, y, z = generate_gp(15, kernel=RBF(beta=1.0), random_state=1, noise=0.1, return_latent=True)
    assert np.array_equal(X, g["X"])
    K = O.cov("rbf", X, None, 1.0, 1.0)[0] + 0.1 * np.eye(15)
    assert relerr(y, sla.cholesky(K, lower=True) @ z) < 1e-12
    X2, y2, z2 = generate_gp(700, n_features=3, noise=0.05, kernel=RBF(sigma=2.0, beta=0.3), random_state=7,
                             return_latent=True)
    K2 = O.cov("rbf", X2, None, 2.0, 0.3)[0] + 0.05 * np.eye(700)
    assert relerr(y2, sla.cholesky(K2, lower=True) @ z2) < 1e-11
    assert generate_gp(5, random_state=0)[1].shape == (5,)


def test_default_noise_stress_case_matches_oracle():
    """SURVEY 8d stress case: the class-default noise variance 0.01 (regression.py:11) on the synthetic
    ARD workload -- cond(Ky) is ~10x the headline's; LML, alpha, gradient, mean and variance still
    within 1e-9 of the oracle (gradient relative to its own scale)."""
    from gaussian_process_amd import ARD, GaussianProcessRegression
    from oracle import gp_oracle as O
    X, y, Xs, beta = O.synthetic_ard(2048, m=128, d=8, seed=2, noise=0.01)
    gp = GaussianProcessRegression(ARD(sigma=1.0, beta=beta))          # sigma = 0.01 default
    gp.fit(X, y)
    lml_o, _, L = O.gpr_lml("ard", X, y, 1.0, beta, 0.01)
    g_o, alpha_o = O.gpr_grad("ard", X, y, 1.0, beta, 0.01, spd_inverse=True)
    mean_o, cov_o = O.gpr_predict("ard", X, y, Xs, 1.0, beta, 0.01, L=L)
    assert abs(gp.log_marginal_likelihood() - lml_o) < 1e-9 * abs(lml_o)
    assert relerr(gp.alpha_(), alpha_o) < 1e-9
    g = gp.grad_log_marginal_likelihood()
    scale = max(np.abs(v).max() for v in g_o.values())
    for k in g_o:
        assert np.abs(g[k] - g_o[k]).max() < 1e-9 * scale
    mean, cov = gp.predict_with_variance(Xs)
    assert relerr(mean, mean_o) < 1e-9 and relerr(cov, cov_o) < 1e-9


def test_failed_evaluation_does_not_leave_a_stale_cache():
    """A non-PD trial theta (an L-BFGS-B excursion) overwrites the resident factor and raises; going
    back to the previous theta must re-evaluate, not serve the cache that described the destroyed
    buffers (regression.py:46-55 recomputes every time, so the reference cannot get this wrong)."""
    from gaussian_process_amd import ARD, GaussianProcessRegression
    X, y, Xs, beta = O.synthetic_ard(500, m=40, d=3, seed=11)
    gp = GaussianProcessRegression(ARD(sigma=1.1, beta=beta), sigma=0.05)
    gp.fit(X, y)
    lml0 = gp.log_marginal_likelihood()
    gp.set_params(sigma=-5.0)                      # K - 5 I is indefinite
    with pytest.raises(np.linalg.LinAlgError):
        gp.grad_log_marginal_likelihood()
    assert gp._state is None
    gp.set_params(sigma=0.05)
    mean_o, cov_o = O.gpr_predict("ard", X, y, Xs, 1.1, beta, 0.05)
    mean, cov = gp.predict_with_variance(Xs)
    assert relerr(mean, mean_o) < TOL and relerr(cov, cov_o) < TOL
    assert gp.log_marginal_likelihood() == lml0
    assert relerr(gp.alpha_(), O.gpr_grad("ard", X, y, 1.1, beta, 0.05)[1]) < TOL


def test_not_positive_definite_returns_before_the_inverse():
    """gpx_gpr_lml_grad reads LAPACK info right after the factorisation: a non-PD theta costs a Gram +
    a (failed) potrf, not the trtri + lauum + traces on a garbage factor (2/3 of the evaluation)."""
    import time
    import torch
    from gaussian_process_amd import ARD, GaussianProcessRegression
    n = 8192
    X, y, _, beta = O.synthetic_ard(n, m=4, d=8, seed=4)
    gp = GaussianProcessRegression(ARD(sigma=1.0, beta=beta), sigma=0.1)
    gp.fit(X, y)
    gp.grad_log_marginal_likelihood()              # warm: buffers allocated, kernels loaded

    def timed(fn, sigma, fails):
        gp.set_params(sigma=sigma)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fails:
            with pytest.raises(np.linalg.LinAlgError):
                fn()
        else:
            fn()
        return time.perf_counter() - t0

    t_grad = min(timed(gp.grad_log_marginal_likelihood, 0.1 + 1e-7 * i, False) for i in (1, 2, 3))
    t_lml = min(timed(gp.log_marginal_likelihood, 0.1 + 1e-7 * i, False) for i in (4, 5, 6))
    # K - 0.5 I fails in an early leading minor, but the factorisation kernels run to the end either way
    t_bad = min(timed(gp.grad_log_marginal_likelihood, -0.5 - 1e-3 * i, True) for i in (1, 2, 3))
    # measured on MI355X at n = 8192: gradient evaluation 20 ms, LML-only (Gram + potrf + 2 solves) 9.9 ms,
    # failed gradient evaluation 9.7 ms (19.4 ms before the early return)
    assert t_bad < 1.1 * t_lml + 1e-3, (t_bad, t_lml, t_grad)
    assert t_bad < 0.7 * t_grad, (t_bad, t_lml, t_grad)


def test_estimator_keeps_one_matrix_for_factor_inverse_factor_and_inverse():
    """LML, gradient and the prediction after a gradient all live in ONE n x n device buffer: W = L^-1 in
    its lower triangle, Ky^-1 (regression.py:60) in the strict upper triangle + a diagonal vector.  The
    estimator's resident device memory is one matrix (+ dinv and vectors), not two."""
    from gaussian_process_amd import ARD, GaussianProcessRegression
    n = 1500
    X, y, Xs, beta = O.synthetic_ard(n, m=64, d=8, seed=7)
    gp = GaussianProcessRegression(ARD(sigma=1.2, beta=beta), sigma=0.07)
    gp.fit(X, y)
    g = gp.grad_log_marginal_likelihood()
    dv = gp._dev
    big = [k for k, v in dv.items() if hasattr(v, "numel") and v.numel() >= n * n]
    assert big == ["Ky"], big
    total = sum(v.numel() * 8 for v in dv.values() if hasattr(v, "numel"))
    ld = dv["Ky"].shape[1]
    assert total < 8 * n * ld + 8 * (128 * 128 * ((n + 127) // 128)) + 8 * 64 * n
    # the buffer's contents: lower triangle W = L^-1, strict upper triangle Ky^-1, diagonal in vec[4n:5n]
    both = dv["rt"].download(dv["Ky"])[:n, :n]
    K = O.cov("ard", X, None, 1.2, beta)[0] + 0.07 * np.eye(n)
    Kinv = np.linalg.inv(K)
    W = np.linalg.inv(np.linalg.cholesky(K))
    assert relerr(np.tril(both), W) < 1e-10
    assert relerr(np.triu(both, 1), np.triu(Kinv, 1)) < 1e-9
    assert relerr(dv["rt"].download(dv["vec"])[4 * n: 5 * n], np.diag(Kinv)) < 1e-9
    # and the values the reference returns
    g_o, _ = O.gpr_grad("ard", X, y, 1.2, beta, 0.07)
    ref = np.concatenate([g_o["sigma"], g_o["cov__sigma"], g_o["cov__beta"]])
    got = np.concatenate([g["sigma"], g["cov__sigma"], g["cov__beta"]])
    assert np.abs(got - ref).max() <= 1e-8 * np.abs(ref).max()
    mean_o, cov_o = O.gpr_predict("ard", X, y, Xs, 1.2, beta, 0.07)
    mean, cov = gp.predict_with_variance(Xs)          # from W in the lower triangle; the upper triangle is not read
    assert relerr(mean, mean_o) < TOL and relerr(cov, cov_o) < TOL


def test_gradient_after_lml_continues_from_the_factor():
    """base.py:35-37 asks for the LML first and the gradient second at the same theta.  The second call goes on
    from the factor and alpha of the first (gpx_gpr_grad_from_factor: no second Gram matrix, no second
    factorisation) and gives the fused evaluation's numbers to rounding: the same factor, inverse and traces; only
    alpha comes from two solves there and from two multiplies by L^-1 in the fused call."""
    import time
    import torch
    from gaussian_process_amd import ARD, GaussianProcessRegression
    n = 3000
    X, y, Xs, beta = O.synthetic_ard(n, m=50, d=8, seed=21)
    fused = GaussianProcessRegression(ARD(sigma=0.9, beta=beta), sigma=0.05)
    fused.fit(X, y)
    g_f = fused.grad_log_marginal_likelihood()            # fused: Gram, potrf, solves, trtri, lauum, traces
    two = GaussianProcessRegression(ARD(sigma=0.9, beta=beta), sigma=0.05)
    two.fit(X, y)
    lml = two.log_marginal_likelihood()
    assert two._state["has_factor"] is True and two._state["grad"] is None
    mean_L = two.predict(Xs)                              # prediction from the factor in between
    g_t = two.grad_log_marginal_likelihood()
    assert two._state["has_factor"] is False and two._state["lml"] == lml
    assert abs(lml - fused.log_marginal_likelihood()) <= 1e-13 * abs(lml)
    g_o, _ = O.gpr_grad("ard", X, y, 0.9, beta, 0.05)
    scale = max(np.abs(v).max() for v in g_o.values())
    for k in g_f:
        assert np.abs(g_f[k] - g_t[k]).max() <= 1e-11 * scale, k
    for k in g_o:
        assert np.abs(g_t[k] - g_o[k]).max() <= 1e-8 * scale, k
    # the buffer now holds what the fused call leaves: prediction from the inverse factor
    mean_o, cov_o = O.gpr_predict("ard", X, y, Xs, 0.9, beta, 0.05)
    mean, cov = two.predict_with_variance(Xs)
    assert relerr(mean_L, mean_o) < TOL and relerr(mean, mean_o) < TOL and relerr(cov, cov_o) < TOL
    # a changed theta never takes the continuation
    two.set_params(sigma=0.06)
    two.log_marginal_likelihood()
    two.set_params(sigma=0.07)
    g7 = two.grad_log_marginal_likelihood()
    fused.set_params(sigma=0.07)
    assert np.array_equal(g7["cov__beta"], fused.grad_log_marginal_likelihood()["cov__beta"])   # both fused
    # cost: LML then gradient is one evaluation, not an LML-only one plus a fused one
    n = 8192
    X, y, _, beta = O.synthetic_ard(n, m=4, d=8, seed=4)
    gp = GaussianProcessRegression(ARD(sigma=1.0, beta=beta), sigma=0.1)
    gp.fit(X, y)
    gp.grad_log_marginal_likelihood()

    def timed(order, sigma):
        gp.set_params(sigma=sigma)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in order:
            f()
        return time.perf_counter() - t0

    t_fused = min(timed([gp.grad_log_marginal_likelihood, gp.log_marginal_likelihood], 0.1 + 1e-7 * i) for i in (1, 2, 3))
    t_two = min(timed([gp.log_marginal_likelihood, gp.grad_log_marginal_likelihood], 0.1 + 1e-7 * i) for i in (4, 5, 6))
    assert t_two < 1.15 * t_fused + 1e-3, (t_two, t_fused)


def test_threads_and_streams_share_the_device_handle_safely():
    """One gpx handle per GPU (shared reduction scratch, pinned staging, stream binding): estimator
    operations are serialised per device by the runtime's lock, and a change of torch stream orders the
    new stream after the old one.  Eight problems evaluated from four threads, each on its own torch
    stream, give the oracle's values."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from gaussian_process_amd import ARD, GaussianProcessRegression
    cases = []
    for i in range(8):
        n = 300 + 37 * i
        X, y, Xs, beta = O.synthetic_ard(n, m=16, d=4, seed=20 + i)
        cases.append((X, y, Xs, beta, 0.05 + 0.01 * i))

    def work(c):
        X, y, Xs, beta, noise = c
        with torch.cuda.stream(torch.cuda.Stream()):
            gp = GaussianProcessRegression(ARD(sigma=1.1, beta=beta), sigma=noise)
            gp.fit(X, y)
            g = gp.grad_log_marginal_likelihood()
            lml = gp.log_marginal_likelihood()
            mean, cov = gp.predict_with_variance(Xs)
        return lml, np.concatenate([g["sigma"], g["cov__sigma"], g["cov__beta"]]), mean, cov

    with ThreadPoolExecutor(4) as ex:
        results = list(ex.map(work, cases))
    for (X, y, Xs, beta, noise), (lml, g, mean, cov) in zip(cases, results):
        g_o, _ = O.gpr_grad("ard", X, y, 1.1, beta, noise)
        ref = np.concatenate([g_o["sigma"], g_o["cov__sigma"], g_o["cov__beta"]])
        assert abs(lml - O.gpr_lml("ard", X, y, 1.1, beta, noise)[0]) <= TOL * abs(lml)
        assert np.abs(g - ref).max() <= 1e-8 * np.abs(ref).max()
        mean_o, cov_o = O.gpr_predict("ard", X, y, Xs, 1.1, beta, noise)
        assert relerr(mean, mean_o) < TOL and relerr(cov, cov_o) < TOL


@pytest.mark.parametrize("seed", range(10))
def test_estimator_against_oracle_on_random_shapes(seed):
    """Seeded random (n, d, m, kernel, noise) through the public estimator against the oracle: odd sizes on both sides
    of every blocking (128-leaves, 256-row trsv leaves, 64/128-wide Gram strips), d from 1 to 12 (compiled and generic
    dimension paths), single test points, both orders of LML / gradient / prediction."""
    from gaussian_process_amd import ARD, RBF, GaussianProcessRegression
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([37, 129, 255, 257, 640, 1023, 1500, 2049, 2600, 3333]))
    d = int(rng.choice([1, 2, 3, 5, 8, 12]))
    m = int(rng.choice([1, 2, 63, 130, 301]))
    kind = "ard" if seed % 2 == 0 else "rbf"
    noise = float(rng.choice([0.02, 0.1, 1.0]))
    sig_k = float(rng.uniform(0.5, 2.0))
    X = rng.uniform(-2, 2, (n, d))
    Xs = rng.uniform(-2, 2, (m, d))
    y = np.sin(X @ rng.uniform(0.3, 1.0, d)) + 0.2 * rng.standard_normal(n)
    beta = rng.uniform(0.2, 2.0, d) if kind == "ard" else float(rng.uniform(0.2, 2.0))
    cov = ARD(sigma=sig_k, beta=beta) if kind == "ard" else RBF(sigma=sig_k, beta=beta)
    gp = GaussianProcessRegression(cov=cov, sigma=noise)
    gp.fit(X, y)
    lml_o, _, L = O.gpr_lml(kind, X, y, sig_k, beta, noise)
    g_o, alpha_o = O.gpr_grad(kind, X, y, sig_k, beta, noise, spd_inverse=True)
    mean_o, cov_o = O.gpr_predict(kind, X, y, Xs, sig_k, beta, noise, L=L)
    scale = max(np.abs(np.atleast_1d(v)).max() for v in g_o.values())

    def check_grad():
        g = gp.grad_log_marginal_likelihood()
        for k in g_o:
            assert np.abs(g[k] - np.atleast_1d(g_o[k])).max() <= 1e-8 * scale, (k, n, d, m, kind)

    def check_predict():
        mean, pc = gp.predict_with_variance(Xs)
        assert mean.shape == (m,) and pc.shape == (m, m)
        assert relerr(mean, mean_o) < TOL and relerr(pc, cov_o) < TOL, (n, d, m, kind)
        assert relerr(gp.predict_with_variance(Xs, full_cov=False)[1], np.diagonal(cov_o)) < 1e-8

    if seed % 3 == 0:            # gradient first: the prediction then runs from the inverse factor
        check_grad()
        check_predict()
    else:
        check_predict()
        check_grad()
    assert abs(gp.log_marginal_likelihood() - lml_o) <= TOL * abs(lml_o)
    assert relerr(gp.alpha_(), alpha_o) < TOL
