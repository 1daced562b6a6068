"""GPU parity of the libgpx primitives, called through the C ABI (ctypes), against NumPy/SciPy
and the oracle on the same seeded inputs.  Tolerances: fp64, relative (Frobenius) <= 1e-12 for
single kernels unless the conditioning of the test problem says otherwise (stated inline)."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

from conftest import relerr
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    from gaussian_process_amd.device import runtime
    r = runtime()
    r.bind_stream()
    return r


def P(t):
    return C.c_void_p(t.data_ptr())


def up(rt, a):
    return rt.upload_matrix(a)


def down(rt, t, r, c):
    return rt.download(t[:r, :c]).copy()


def ok(rc):
    from gaussian_process_amd import _lib
    _lib.check(rc, "gpx call")


def spd(n, seed, cond_shift=1.0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n + 3))
    return A @ A.T / n + cond_shift * np.eye(n)


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (128, 128, 16), (130, 70, 33), (257, 300, 129), (64, 1, 200)])
def test_gemm_layouts(rt, ta, tb, M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K + ta * 2 + tb)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    Cm = rng.standard_normal((M, N))
    dA, dB, dC = up(rt, A), up(rt, B), up(rt, Cm)
    ok(rt.lib.gpx_gemm(rt.h, ta, tb, M, N, K, 0.75, P(dA), dA.shape[1], P(dB), dB.shape[1], -0.5,
                       P(dC), dC.shape[1]))
    ref = 0.75 * ((A.T if ta else A) @ (B.T if tb else B)) - 0.5 * Cm
    assert relerr(down(rt, dC, M, N), ref) < 1e-13


def test_gemm_identity_asymmetric(rt):
    """A = I with an asymmetric B pins the MFMA C/D lane map (row/col not swapped)."""
    n = 128
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)
    dA, dB, dC = up(rt, np.eye(n)), up(rt, B), up(rt, np.zeros((n, n)))
    ok(rt.lib.gpx_gemm(rt.h, 0, 0, n, n, n, 1.0, P(dA), dA.shape[1], P(dB), dB.shape[1], 0.0, P(dC),
                       dC.shape[1]))
    assert np.array_equal(down(rt, dC, n, n), B)


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 1, 17408, 512, 40), (0, 0, 20480, 512, 24), (1, 1, 17000, 500, 17),
                                         (0, 1, 16512, 16512, 8)])
def test_gemm_multi_round_launches(rt, ta, tb, M, N, K):
    """More than 512 tiles: the XCD super-tile mapping, and for the first three shapes a sparse last
    round (544 / 640 / 532 tiles of 128 x 128) that is launched separately at one workgroup per CU."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    Cm = rng.standard_normal((M, N))
    dA, dB, dC = up(rt, A), up(rt, B), up(rt, Cm)
    ok(rt.lib.gpx_gemm(rt.h, ta, tb, M, N, K, -1.0, P(dA), dA.shape[1], P(dB), dB.shape[1], 1.0,
                       P(dC), dC.shape[1]))
    ref = Cm - (A.T if ta else A) @ (B.T if tb else B)
    got = down(rt, dC, M, N)
    assert relerr(got, ref) < 1e-13
    assert np.max(np.abs(got - ref)) < 1e-12 * max(1.0, np.max(np.abs(ref)))   # no tile skipped or done twice


@pytest.mark.parametrize("ta,tb", [(0, 1), (1, 0)])
def test_gemm_very_long_k_runs_in_chunks(rt, ta, tb):
    """K >= 16384 with plain operands is issued as consecutive launches of K = 8192 (the last one takes
    the remainder, here 8192 + 70): same product, C read once with beta and re-read with 1 afterwards,
    for k-contiguous and row-contiguous operands.  The host product is the checker (NumPy, fp64)."""
    M, N, K = 4096, 4096, 3 * 8192 + 70
    rng = np.random.default_rng(K + ta)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    Cm = rng.standard_normal((M, N))
    dA, dB, dC = up(rt, A), up(rt, B), up(rt, Cm)
    ok(rt.lib.gpx_gemm(rt.h, ta, tb, M, N, K, -0.5, P(dA), dA.shape[1], P(dB), dB.shape[1], 0.25,
                       P(dC), dC.shape[1]))
    ref = 0.25 * Cm - 0.5 * ((A.T if ta else A) @ (B.T if tb else B))
    assert relerr(down(rt, dC, M, N), ref) < 1e-13


@pytest.mark.parametrize("nb,ncols,K,M_extra", [(128, 5, 40, 0), (256, 7, 272, 100), (512, 12, 2048, 37)])
def test_gemm_colblocks_matches_numpy(rt, nb, ncols, K, M_extra):
    """The grouped trailing update of the sharded Cholesky: ONE launch over several nb-wide column blocks, each
    landing in its own slab (leading dimension nb) that starts at its own first row; tiles above a slab's first
    row are skipped and the slabs' other contents are untouched.  Checked against NumPy block by block."""
    import torch
    rng = np.random.default_rng(nb + ncols)
    # local rows: the slabs start at rows 0, nb, 2 nb, ... of a row space of ncols * nb (+ a ragged piece) rows
    R = ncols * nb - (nb - M_extra if M_extra else 0)
    row0 = np.array([c * nb for c in range(ncols)], dtype=np.int32)
    w_last = 70 if M_extra else nb                               # ragged last column block
    N = (ncols - 1) * nb + w_last
    heights = [(R - row0[c] + nb - 1) // nb * nb for c in range(ncols)]       # whole tiles high
    offs, off = [], 64                                           # the store starts with an unrelated 64 doubles
    for c in range(ncols):
        offs.append(off - int(row0[c]) * nb)
        off += heights[c] * nb
    store_h = rng.standard_normal(off + 32)
    A = rng.standard_normal((R, K))
    B = rng.standard_normal((N, K))
    dA, dB = up(rt, A), up(rt, B)
    store = rt.torch.from_numpy(store_h.copy()).to(rt.dev)
    d_off = rt.torch.tensor(offs, dtype=rt.torch.int64, device=rt.dev)
    # bit 0 of a table entry: the slab starts with the column's diagonal tile (here: the even blocks)
    d_row0 = rt.torch.from_numpy(row0 | (1 - np.arange(ncols, dtype=np.int32) % 2)).to(rt.dev)
    first = 2 if ncols > 4 else 0                                # launch over column blocks first.. : row base = its first row
    rb = int(row0[first])
    Nl = N - first * nb
    ok(rt.lib.gpx_gemm_colblocks(rt.h, R - rb, Nl, K, -1.0, C.c_void_p(dA.data_ptr() + 8 * rb * dA.shape[1]), dA.shape[1],
                                 C.c_void_p(dB.data_ptr() + 8 * first * nb * dB.shape[1]), dB.shape[1], 1.0, P(store), nb, rb,
                                 C.c_void_p(d_off.data_ptr() + 8 * first), C.c_void_p(d_row0.data_ptr() + 4 * first)))
    rt.torch.cuda.synchronize()
    got = store.cpu().numpy()
    want = store_h.copy()

    def live(c, i, w):
        """columns of row i of block c that the launch writes: inside the slab's first nb rows (its diagonal tile)
        the 128 x 128 tiles strictly above the diagonal are skipped"""
        rel = i - int(row0[c])
        return min(w, (rel // 128 + 1) * 128) if (rel < nb and c % 2 == 0) else w

    for c in range(first, ncols):
        w = nb if c < ncols - 1 else w_last
        prod = A[row0[c]:] @ B[c * nb: c * nb + w].T
        for i in range(row0[c], R):
            o = offs[c] + i * nb
            wl = live(c, i, w)
            want[o: o + wl] -= prod[i - row0[c]][:wl]
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-13 * scale * max(1.0, K ** 0.5)
    assert nb == 128 or any(live(c, int(row0[c]), nb) < nb for c in range(first, ncols))
    untouched = np.ones(len(want), dtype=bool)
    for c in range(first, ncols):
        w = nb if c < ncols - 1 else w_last
        for i in range(row0[c], R):
            untouched[offs[c] + i * nb: offs[c] + i * nb + live(c, i, w)] = False
    assert np.array_equal(got[untouched], store_h[untouched])


@pytest.mark.parametrize("m,n", [(37, 128), (900, 512), (3000, 2048), (130, 300)])
def test_trmm_right_matches_numpy(rt, m, n):
    """Out = X W^T for a lower-triangular W whose strict upper triangle holds garbage (NaN): only j <= i is read.
    With W = L_kk^-1 this is the panel solve A21 L11^-T of the sharded Cholesky as one product."""
    rng = np.random.default_rng(m + n)
    W = np.tril(rng.standard_normal((n, n)))
    X = rng.standard_normal((m, n))
    Wd = W.copy()
    Wd[np.triu_indices(n, 1)] = np.nan
    dW, dX = up(rt, Wd), up(rt, X)
    Out = rt.matrix(m, n)
    ok(rt.lib.gpx_trmm_right(rt.h, m, n, P(dW), dW.shape[1], P(dX), dX.shape[1], P(Out), Out.shape[1]))
    assert relerr(down(rt, Out, m, n), X @ W.T) < 1e-13


# ----------------------------------------------------------------------------- Gram
@pytest.mark.parametrize("kind,d", [("rbf", 1), ("rbf", 3), ("ard", 2), ("ard", 8), ("ard", 5)])
@pytest.mark.parametrize("n1,n2", [(1, 1), (15, 50), (129, 257), (300, 300), (700, 1537), (1100, 512)])   # the last two: 512-column strips, several row blocks of the packed lower grid
def test_gram_matches_oracle(rt, kind, d, n1, n2):
    rng = np.random.default_rng(n1 + 7 * n2 + d)
    X1 = rng.uniform(-3, 3, (n1, d))
    X2 = rng.uniform(-3, 3, (n2, d))
    beta = rng.uniform(0.05, 2.0, d if kind == "ard" else 1)
    sigma = 1.7
    from gaussian_process_amd import _lib
    barr, bp = _lib.host_doubles(beta)
    dX1, dX2 = up(rt, X1), up(rt, X2)
    K = rt.matrix(n1, n2)
    ok(rt.lib.gpx_gram(rt.h, _lib.GPX_KIND[kind], P(dX1), n1, dX1.shape[1], P(dX2), n2, dX2.shape[1], d,
                       sigma, bp, 0.0, P(K), K.shape[1], 0))
    Ko, _, _ = O.cov(kind, X1, X2, sigma, beta)
    assert relerr(down(rt, K, n1, n2), Ko) < 1e-13
    # square, lower, with the noise on the diagonal
    Kl = rt.matrix(n1, n1)
    Kl.fill_(float("nan"))
    ok(rt.lib.gpx_gram(rt.h, _lib.GPX_KIND[kind], P(dX1), n1, dX1.shape[1], None, 0, 0, d, sigma, bp,
                       0.25, P(Kl), Kl.shape[1], 1))
    Ks, _, _ = O.cov(kind, X1, None, sigma, beta)
    got = down(rt, Kl, n1, n1)
    assert relerr(np.tril(got), np.tril(Ks + 0.25 * np.eye(n1))) < 1e-13
    # inside each 128-wide diagonal block the strict upper triangle is zero
    for b0 in range(0, n1, 128):
        blk = got[b0:b0 + 128, b0:b0 + 128]
        assert np.all(np.triu(blk, 1) == 0.0)


@pytest.mark.parametrize("kind", ["rbf", "ard"])
def test_gram_huge_exponent_is_an_exact_zero(rt, kind):
    """beta * dist^2 ~ 1e9 .. 1e300 (unscaled features, a large beta on an L-BFGS-B excursion): exp underflows to an
    exact 0 as numpy's does (kernels.py:37,79) -- no NaN, no garbage from an overflowing int conversion."""
    from gaussian_process_amd import _lib
    d = 2
    X = np.array([[0.0, 0.0], [1.0, 0.0], [3e4, 0.0], [0.0, 2e7], [1e150, 0.0], [1.0, 1e-3]])
    beta = np.array([2.0, 1.0]) if kind == "ard" else np.array([2.0])
    barr, bp = _lib.host_doubles(beta)
    dX = up(rt, X)
    n = X.shape[0]
    K = rt.matrix(n, n)
    ok(rt.lib.gpx_gram(rt.h, _lib.GPX_KIND[kind], P(dX), n, dX.shape[1], None, 0, 0, d, 1.5, bp, 0.0, P(K), K.shape[1], 0))
    got = down(rt, K, n, n)
    with np.errstate(over="ignore"):
        Ko, _, _ = O.cov(kind, X, None, 1.5, beta)
    assert np.isfinite(got).all()
    assert np.array_equal(got == 0.0, Ko == 0.0)
    assert relerr(got, Ko) < 1e-13
    # the fused gradient trace uses the same exp: finite sums with such points in the data
    alpha = up(rt, np.ones((1, n)))
    Kinv = up(rt, np.eye(n))
    out = (C.c_double * (2 + len(beta)))()
    ok(rt.lib.gpx_grad_trace(rt.h, _lib.GPX_KIND[kind], P(dX), n, dX.shape[1], d, 1.5, bp, P(alpha), P(Kinv),
                             Kinv.shape[1], None, C.cast(out, C.POINTER(C.c_double))))
    assert np.isfinite(np.array(list(out))[:2]).all()


def test_non_finite_inputs_behave_as_in_the_reference(rt):
    """NumPy's exp propagates NaN (kernels.py:37,79); the device exp clamps with a min that drops it, so the
    Python layer puts the NaNs back.  predict raises as scipy's solve_triangular does (regression.py:38)."""
    from gaussian_process_amd import RBF, ARD, GaussianProcessRegression
    rng = np.random.default_rng(0)
    X1, X2 = rng.uniform(-1, 1, (6, 2)), rng.uniform(-1, 1, (5, 2))
    X2[3, 1] = np.nan
    for cov in (RBF(1.0, 0.5), ARD(1.0, [0.5, 2.0])):
        kv = cov(X1, X2)
        assert np.isnan(kv.K[:, 3]).all() and np.isfinite(np.delete(kv.K, 3, axis=1)).all()
        assert np.isnan(kv.dtheta["sigma"][:, 3]).all() and np.isnan(kv.dtheta["beta"][..., 3]).all()
    gp = GaussianProcessRegression(cov=RBF(), sigma=0.1)
    gp.fit(X1, rng.standard_normal(6))
    with pytest.raises(ValueError):
        gp.predict(X2)


def test_gram_dtheta_matches_oracle(rt):
    from gaussian_process_amd import RBF, ARD
    rng = np.random.default_rng(5)
    X1, X2 = rng.uniform(-3, 3, (70, 3)), rng.uniform(-3, 3, (45, 3))
    kv = ARD(sigma=1.3, beta=[0.2, 1.0, 2.5])(X1, X2)
    K, ds, db = O.ard_cov(X1, X2, 1.3, np.array([0.2, 1.0, 2.5]), want_grad=True)
    assert relerr(kv.K, K) < 1e-13 and relerr(kv.dtheta["sigma"], ds) < 1e-13
    assert kv.dtheta["beta"].shape == (3, 70, 45) and relerr(kv.dtheta["beta"], db) < 1e-13
    kv = RBF(sigma=0.7, beta=0.4)(X1)
    K, ds, db = O.rbf_cov(X1, None, 0.7, 0.4, want_grad=True)
    assert relerr(kv.K, K) < 1e-13 and relerr(kv.dtheta["sigma"], ds) < 1e-13
    assert kv.dtheta["beta"].shape == (70, 70)
    assert np.abs(kv.dtheta["beta"] - db).max() < 1e-13 * np.abs(db).max()
    assert list(kv.dtheta) == ["sigma", "beta"]
    # the planes are built on first read, from the parameters the kernel had when it was CALLED
    cov = ARD(sigma=1.3, beta=[0.2, 1.0, 2.5])
    kv = cov(X1, X2)
    assert "pending" in repr(kv.dtheta) and len(kv.dtheta) == 2 and set(kv.dtheta.keys()) == {"sigma", "beta"}
    cov.set_params(beta=np.array([9.0, 9.0, 9.0]), sigma=5.0)
    K, ds, db = O.ard_cov(X1, X2, 1.3, np.array([0.2, 1.0, 2.5]), want_grad=True)
    assert relerr(kv.dtheta["beta"], db) < 1e-13 and "materialised" in repr(kv.dtheta)
    assert relerr(dict(kv.dtheta.items())["sigma"], ds) < 1e-13
    with pytest.raises(KeyError):
        kv.dtheta["gamma"]


@pytest.mark.parametrize("n", [8192, 9000, 10241])
def test_single_rhs_solves_through_block_inverses(rt, n):
    """From n = 8192 on a single-right-hand-side solve goes through explicit inverses of the 1024 x 1024 diagonal
    blocks (trsv.hip: 2 streaming launches per block instead of a chain of 256-row leaves); the ragged tail
    (n mod 1024 rows: 808, 1) goes through the recursion.  Against SciPy's solve_triangular on the same factor
    (regression.py:37,50 and the second half of cho_solve behind :61), strided and contiguous vectors."""
    rng = np.random.default_rng(n)
    L = np.tril(rng.uniform(-1, 1, (n, n))) / np.sqrt(n)
    L[np.arange(n), np.arange(n)] = rng.uniform(1.0, 2.0, n)
    dL = up(rt, L)
    ld = dL.shape[1]
    dinv = rt.dinv(n)
    ok(rt.lib.gpx_diag_inv(rt.h, n, P(dL), ld, P(dinv)))
    b = rng.standard_normal(n)
    for trans in (0, 1):
        want = sla.solve_triangular(L, b, lower=True, trans="T" if trans else "N", check_finite=False)
        for inc in (1, 2):
            x = rt.torch.zeros((n, inc), dtype=rt.torch.float64, device=rt.dev)
            x[:, 0].copy_(rt.torch.from_numpy(b))
            ok(rt.lib.gpx_trsm(rt.h, trans, n, 1, P(dL), ld, P(dinv), P(x), inc))
            got = rt.download(x[:, 0])
            assert relerr(got, want) < 1e-12, (trans, inc)
    # the factor itself is untouched (the block inverses are built on copies)
    assert np.array_equal(down(rt, dL, n, n), L)


# ----------------------------------------------------------------------------- Cholesky family
@pytest.mark.parametrize("n", [1, 2, 15, 100, 128, 129, 255, 256, 300, 1000, 1024, 2050, 4096, 4100, 5000, 8192])   # 256 / 1024 / 4096 / 8192: the inversion goes level by level (batched launches)
def test_potrf_trsm_trtri_lauum(rt, n):
    A = spd(n, n)
    dA = up(rt, A)
    dinv = rt.dinv(n)
    ld = dA.shape[1]
    ok(rt.lib.gpx_potrf(rt.h, n, P(dA), ld, P(dinv)))
    L = np.tril(down(rt, dA, n, n))
    Lref = sla.cholesky(A, lower=True)
    assert relerr(L, Lref) < 1e-13
    assert relerr(L @ L.T, A) < 1e-14
    ldet = C.c_double()
    ok(rt.lib.gpx_logdet(rt.h, n, P(dA), ld, C.byref(ldet)))
    assert abs(ldet.value - np.log(np.diag(Lref)).sum()) <= 1e-12 * max(1.0, abs(ldet.value))
    # dinv from scratch equals the one potrf produced
    dinv2 = rt.dinv(n)
    ok(rt.lib.gpx_diag_inv(rt.h, n, P(dA), ld, P(dinv2)))
    assert relerr(rt.download(dinv2), rt.download(dinv)) < 1e-13
    rng = np.random.default_rng(n)
    for nrhs in (1, 3, 200):
        B = rng.standard_normal((n, nrhs))
        for trans in (0, 1):
            dB = up(rt, B)
            ok(rt.lib.gpx_trsm(rt.h, trans, n, nrhs, P(dA), ld, P(dinv), P(dB), dB.shape[1]))
            ref = sla.solve_triangular(Lref, B, lower=True, trans="T" if trans else "N")
            assert relerr(down(rt, dB, n, nrhs), ref) < 1e-12
    # potri = trtri + lauum
    ok(rt.lib.gpx_trtri(rt.h, n, P(dA), ld, P(dinv)))      # scratch: the blocks above the diagonal blocks
    W = np.tril(down(rt, dA, n, n))
    assert relerr(W, np.linalg.inv(Lref)) < 1e-12
    Kinv = rt.matrix(n, n)
    ok(rt.lib.gpx_lauum(rt.h, n, P(dA), ld, P(Kinv), Kinv.shape[1]))
    assert relerr(np.tril(down(rt, Kinv, n, n)), np.tril(np.linalg.inv(A))) < 1e-12
    ok(rt.lib.gpx_symmetrize(rt.h, n, P(Kinv), Kinv.shape[1]))
    assert relerr(down(rt, Kinv, n, n), np.linalg.inv(A)) < 1e-12
    # the same product inside W's own buffer: strict upper triangle + diagonal vector, W untouched
    kd = rt.vector(n)
    ok(rt.lib.gpx_lauum_upper(rt.h, n, P(dA), ld, P(kd)))
    both = down(rt, dA, n, n)
    Ainv = np.linalg.inv(A)
    assert np.array_equal(np.tril(both), W)
    assert relerr(np.triu(both, 1), np.triu(Ainv, 1)) < 1e-12
    assert relerr(rt.download(kd)[:n], np.diag(Ainv)) < 1e-12


@pytest.mark.parametrize("n", [100, 257, 513, 1024, 1153, 4097, 9000, 16500])
def test_trsv_single_rhs(rt, n):
    """One right-hand side goes through the streaming TRSV (leaf + GEMV recursion), both directions,
    including strided vectors; the factor here is a generic lower-triangular matrix, not a Cholesky
    factor (solve_triangular at regression.py:37,50)."""
    rng = np.random.default_rng(n)
    L = np.tril(rng.standard_normal((n, n)) / np.sqrt(n))
    L[np.diag_indices(n)] = 1.0 + rng.random(n)
    dL, dinv = up(rt, L), rt.dinv(n)
    ok(rt.lib.gpx_diag_inv(rt.h, n, P(dL), dL.shape[1], P(dinv)))
    b = rng.standard_normal((n, 1))
    for trans in (0, 1):
        ref = sla.solve_triangular(L, b, lower=True, trans="T" if trans else "N")
        for ldb in (32, 2):
            import torch
            dB = torch.zeros((n, ldb), dtype=torch.float64, device=rt.dev)
            dB[:, :1].copy_(torch.from_numpy(b))
            ok(rt.lib.gpx_trsm(rt.h, trans, n, 1, P(dL), dL.shape[1], P(dinv), P(dB), ldb))
            got = rt.download(dB)
            assert relerr(got[:, :1], ref) < 1e-12
            assert not got[:, 1:].any()          # neighbours of the strided vector untouched


@pytest.mark.parametrize("n,bad", [(10, 4), (300, 0), (300, 131), (300, 299)])
def test_potrf_reports_leading_minor(rt, n, bad):
    """LAPACK info semantics: first leading minor that is not positive definite (regression.py:36)."""
    A = spd(n, 3)
    A[bad, bad] = -1.0
    dA = up(rt, A)
    rc = rt.lib.gpx_potrf(rt.h, n, P(dA), dA.shape[1], P(rt.dinv(n)))
    with pytest.raises(np.linalg.LinAlgError):
        sla.cholesky(A, lower=True)
    assert rc == bad + 1
    from gaussian_process_amd import _lib
    with pytest.raises(np.linalg.LinAlgError, match="%d-th leading minor" % (bad + 1)):
        _lib.check(rc, "potrf")


@pytest.mark.parametrize("n,m", [(128, 64), (300, 517), (1000, 33), (1, 5)])
def test_potrf_async_and_trsm_right(rt, n, m):
    """The panel step of a blocked Cholesky as the multi-GPU driver issues it: factor a diagonal
    block without a host sync (info stays on the device), then X <- X L^-T for the rows below."""
    import torch
    A = spd(n, 11)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((m, n))
    dA, dX, dinv = up(rt, A), up(rt, X), rt.dinv(n)
    info = torch.zeros(4, dtype=torch.int32, device=rt.dev)
    ok(rt.lib.gpx_potrf_async(rt.h, n, P(dA), dA.shape[1], P(dinv), P(info), 1000))
    ok(rt.lib.gpx_trsm_right(rt.h, m, n, P(dA), dA.shape[1], P(dinv), P(dX), dX.shape[1]))
    assert int(info[0].item()) == 0
    L = sla.cholesky(A, lower=True)
    assert relerr(np.tril(down(rt, dA, n, n)), L) < 1e-13
    want = sla.solve_triangular(L, X.T, lower=True).T
    assert relerr(down(rt, dX, m, n), want) < 1e-11
    # failure: first bad minor, offset by the caller's global row offset; later failures don't overwrite
    B = spd(max(n, 2), 12)
    nb = B.shape[0]
    B[nb - 1, nb - 1] = -1.0
    dB = up(rt, B)
    ok(rt.lib.gpx_potrf_async(rt.h, nb, P(dB), dB.shape[1], P(rt.dinv(nb)), P(info), 1000))
    dB2 = up(rt, -np.eye(nb))
    ok(rt.lib.gpx_potrf_async(rt.h, nb, P(dB2), dB2.shape[1], P(rt.dinv(nb)), P(info), 5000))
    assert int(info[0].item()) == 1000 + nb


def test_bad_arguments_are_rejected_not_faulted(rt):
    from gaussian_process_amd import _lib
    A = up(rt, spd(8, 1))
    assert rt.lib.gpx_potrf(rt.h, 8, P(A), 7, P(rt.dinv(8))) == -1      # odd / too small ld
    assert rt.lib.gpx_potrf(rt.h, 8, C.c_void_p(A.data_ptr() + 8), 32, P(rt.dinv(8))) == -1  # misaligned
    assert rt.lib.gpx_potrf(rt.h, 8, None, 32, P(rt.dinv(8))) == -1
    assert "potrf" in _lib.last_error()
    assert rt.lib.gpx_potrf(rt.h, 0, None, 0, None) == 0                  # empty input is fine


@pytest.mark.parametrize("n,m", [(300, 129), (1000, 64), (5, 3)])
def test_syrk_sub_gemv(rt, n, m):
    rng = np.random.default_rng(n + m)
    V = rng.standard_normal((n, m))
    Cm = spd(m, 2)
    dV, dC = up(rt, V), up(rt, Cm)
    ok(rt.lib.gpx_syrk_sub(rt.h, n, m, P(dV), dV.shape[1], P(dC), dC.shape[1], 0.3, 0))
    assert relerr(down(rt, dC, m, m), Cm - V.T @ V + 0.3 * np.eye(m)) < 1e-13
    x = rng.standard_normal(n)
    dx, dy = rt.upload_vector(x), rt.vector(m)
    ok(rt.lib.gpx_gemv(rt.h, 1, n, m, P(dV), dV.shape[1], P(dx), P(dy)))
    assert relerr(rt.download(dy[:m]), V.T @ x) < 1e-13
    z = rng.standard_normal(m)
    dz, dw = rt.upload_vector(z), rt.vector(n)
    ok(rt.lib.gpx_gemv(rt.h, 0, n, m, P(dV), dV.shape[1], P(dz), P(dw)))
    assert relerr(rt.download(dw[:n]), V @ z) < 1e-13


@pytest.mark.parametrize("kind,d,n", [("ard", 8, 300), ("rbf", 2, 129), ("ard", 3, 64), ("ard", 8, 1500), ("ard", 32, 700),
                                      ("ard", 5, 1100), ("rbf", 8, 1030)])
def test_grad_trace_matches_oracle(rt, kind, d, n):
    """d = 32 is GPX_MAXD (generic-dimension kernel, 4 row sub-tiles per workgroup); n = 1500 / 1100 / 1030 span
    several 512-row (256-row) workgroup strips with ragged edges."""
    from gaussian_process_amd import _lib
    rng = np.random.default_rng(n)
    X = rng.uniform(-3, 3, (n, d))
    y = rng.standard_normal(n)
    beta = rng.uniform(0.05, 1.0, d if kind == "ard" else 1)
    sigma, noise = 1.4, 0.1
    g_o, alpha = O.gpr_grad(kind, X, y, sigma, beta, noise)
    K, _, _ = O.cov(kind, X, None, sigma, beta)
    Kinv = np.linalg.inv(K + noise * np.eye(n))
    dX, dal, dKi = up(rt, X), rt.upload_vector(alpha), up(rt, np.tril(Kinv))
    barr, bp = _lib.host_doubles(beta)
    out = (C.c_double * (2 + len(barr)))()
    ok(rt.lib.gpx_grad_trace(rt.h, _lib.GPX_KIND[kind], P(dX), n, dX.shape[1], d, sigma, bp, P(dal),
                             P(dKi), dKi.shape[1], None, C.cast(out, C.POINTER(C.c_double))))
    got = np.array(list(out))
    ref = np.concatenate([g_o["sigma"], g_o["cov__sigma"], g_o["cov__beta"]])
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()
    # upper storage (what gpx_lauum_upper leaves): strict upper triangle + diagonal vector; the lower
    # triangle of the buffer belongs to W and must never be read -- NaN there must not reach the sums
    Ku = np.triu(Kinv, 1) + np.tril(np.full((n, n), np.nan))
    dKu, dkd = up(rt, Ku), rt.upload_vector(np.diag(Kinv).copy())
    out2 = (C.c_double * (2 + len(barr)))()
    ok(rt.lib.gpx_grad_trace(rt.h, _lib.GPX_KIND[kind], P(dX), n, dX.shape[1], d, sigma, bp, P(dal),
                             P(dKu), dKu.shape[1], P(dkd), C.cast(out2, C.POINTER(C.c_double))))
    got2 = np.array(list(out2))
    assert np.abs(got2 - ref).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("kind,d,n,cuts", [("ard", 8, 300, [0, 128, 256, 300]), ("rbf", 2, 257, [0, 128, 256, 257]),
                                           ("ard", 3, 200, [0, 64, 200])])
def test_grad_trace_slabs_add_up(rt, kind, d, n, cuts):
    """Block-column slabs of the lower triangle (rows from the slab's first column down, as the
    multi-GPU owners hold them) sum to the whole-matrix gradient of regression.py:62-75."""
    from gaussian_process_amd import _lib
    rng = np.random.default_rng(n + 1)
    X = rng.uniform(-3, 3, (n, d))
    y = rng.standard_normal(n)
    beta = rng.uniform(0.05, 1.0, d if kind == "ard" else 1)
    sigma, noise = 0.9, 0.2
    g_o, alpha = O.gpr_grad(kind, X, y, sigma, beta, noise)
    K, _, _ = O.cov(kind, X, None, sigma, beta)
    Kinv = np.linalg.inv(K + noise * np.eye(n))
    dX, dal = up(rt, X), rt.upload_vector(alpha)
    barr, bp = _lib.host_doubles(beta)
    total = np.zeros(2 + len(barr))
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        slab = up(rt, Kinv[c0:, c0:c1])               # entries above the diagonal present but must be ignored
        out = (C.c_double * (2 + len(barr)))()
        ok(rt.lib.gpx_grad_trace_slab(rt.h, _lib.GPX_KIND[kind], P(dX), n, dX.shape[1], d, sigma, bp, P(dal),
                                      P(slab), slab.shape[1], c0, c0, n - c0, c1 - c0,
                                      C.cast(out, C.POINTER(C.c_double))))
        total += np.array(list(out))
    ref = np.concatenate([g_o["sigma"], g_o["cov__sigma"], g_o["cov__beta"]])
    assert np.abs(total - ref).max() <= 1e-10 * np.abs(ref).max()
    # a slab outside the matrix is rejected, not faulted
    assert rt.lib.gpx_grad_trace_slab(rt.h, _lib.GPX_KIND[kind], P(dX), n, dX.shape[1], d, sigma, bp, P(dal),
                                      P(slab), slab.shape[1], n - 10, 0, 64, 64,
                                      C.cast(out, C.POINTER(C.c_double))) == -1


@pytest.mark.parametrize("n,m", [(1, 1), (130, 3), (700, 64), (1025, 1), (2048, 700)])
def test_trmm_lower_times_matrix(rt, n, m):
    """Out = L B and Out = L^T B reading only the lower triangle of L (garbage above it must not matter)."""
    rng = np.random.default_rng(n + m)
    L = np.tril(rng.standard_normal((n, n)))
    B = rng.standard_normal((n, m))
    dL = up(rt, L + np.triu(np.full((n, n), np.nan), 1))     # NaN above the diagonal
    dB, dO = up(rt, B), rt.matrix(n, m)
    ok(rt.lib.gpx_trmm(rt.h, 0, n, m, P(dL), dL.shape[1], P(dB), dB.shape[1], P(dO), dO.shape[1]))
    assert relerr(down(rt, dO, n, m), L @ B) < 1e-13
    ok(rt.lib.gpx_trmm(rt.h, 1, n, m, P(dL), dL.shape[1], P(dB), dB.shape[1], P(dO), dO.shape[1]))
    assert relerr(down(rt, dO, n, m), L.T @ B) < 1e-13


@pytest.mark.parametrize("n", [1, 2, 100, 128, 129, 1000, 2047, 4096, 5001, 16500])
def test_trmv_lower_times_vector(rt, n):
    """y = W x and y = W^T x reading only the lower triangle of W (NaN above it): the streaming replacement of the
    two vector solves once the gradient pass holds W = L^-1."""
    rng = np.random.default_rng(n)
    W = np.tril(rng.standard_normal((n, n)))
    x = rng.standard_normal(n)
    dW = up(rt, W + np.triu(np.full((n, n), np.nan), 1))
    dx, dy = rt.upload_vector(x), rt.vector(n)
    ok(rt.lib.gpx_trmv(rt.h, 0, n, P(dW), dW.shape[1], P(dx), P(dy)))
    assert relerr(rt.download(dy[:n]), W @ x) < 1e-13
    ok(rt.lib.gpx_trmv(rt.h, 1, n, P(dW), dW.shape[1], P(dx), P(dy)))
    assert relerr(rt.download(dy[:n]), W.T @ x) < 1e-13
    with pytest.raises(Exception):
        ok(rt.lib.gpx_trmv(rt.h, 0, n, P(dW), dW.shape[1], P(dx), P(dx)))


def test_trmv_transposed_column_bands_give_the_same_product():
    """y = W^T x keeps n / rows chunks of partial column sums in the handle's scratch; beyond n ~ 131 000 they no
    longer fit at once and the columns go band by band (found at n = 131072: tools/big_n_grad.py).  Forced here with
    bands of 256 columns (development override read at first use, hence the child process): bitwise the same sums."""
    import os
    import subprocess
    import sys
    code = """
import numpy as np, sys
sys.path.insert(0, %r)
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime, ptr
rt = runtime(); rt.bind_stream()
rng = np.random.default_rng(11)
n = 1337
W = np.tril(rng.standard_normal((n, n))); x = rng.standard_normal(n)
dW, dx, dy = rt.upload_matrix(W), rt.upload_vector(x), rt.vector(n)
_lib.check(rt.lib.gpx_trmv(rt.h, 1, n, ptr(dW), dW.shape[1], ptr(dx), ptr(dy)), "trmv")
y = rt.download(dy[:n])
print(repr(float(np.abs(y - W.T @ x).max() / np.abs(W.T @ x).max())), " ".join(repr(v) for v in y[:: 97]))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for band in ("", "256"):
        env = dict(os.environ)
        env.pop("GPX_TRMV_BAND_COLS", None)
        if band:
            env["GPX_TRMV_BAND_COLS"] = band
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        line = r.stdout.strip().splitlines()[-1].split()
        assert float(line[0]) < 1e-13, line[0]
        outs.append(line[1:])
    assert outs[0] == outs[1]


def test_grad_trace_row_bands_give_the_same_sums():
    """When the per-workgroup partials of a large n x d would not fit the reduction scratch the launch is
    split into row bands whose sums accumulate; forced here with one block row per band (development
    override read at first use, hence the child process)."""
    import os
    import subprocess
    import sys
    code = """
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime, ptr
rt = runtime(); rt.bind_stream()
rng = np.random.default_rng(5)
n, d = 1700, 8
X = rng.uniform(-3, 3, (n, d)); al = rng.standard_normal(n)
M = rng.standard_normal((n, n)); M = np.tril(M @ M.T / n)
beta = rng.uniform(0.05, 1.0, d)
dX, dal, dM = rt.upload_matrix(X), rt.upload_vector(al), rt.upload_matrix(M)
barr, bp = _lib.host_doubles(beta)
out = (C.c_double * (2 + d))()
_lib.check(rt.lib.gpx_grad_trace(rt.h, 1, ptr(dX), n, dX.shape[1], d, 1.3, bp, ptr(dal), ptr(dM), dM.shape[1],
                                 None, C.cast(out, C.POINTER(C.c_double))), "grad_trace")
print(" ".join(repr(v) for v in out))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for band in ("", "1"):
        env = dict(os.environ)
        env.pop("GPX_TRACE_BAND_ROWS", None)
        if band:
            env["GPX_TRACE_BAND_ROWS"] = band
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.array([float(v) for v in r.stdout.strip().splitlines()[-1].split()]))
    assert np.abs(outs[0] - outs[1]).max() <= 1e-12 * np.abs(outs[0]).max()


@pytest.mark.parametrize("m,k,lower", [(4096, 1024, 1), (4200, 1100, 1), (4096, 2048, 0), (5000, 1500, 1)])
def test_gemm_mid_size_lower_launches_packed_ids_and_split_k_tail(rt, m, k, lower):
    """Lower launches of 513..4096 tiles use packed tile ids; a short last round (<= 128 tiles) with a long
    K is computed split over K into the handle's workspace and summed in a fixed order (gemm_f64.hip,
    launch_cfg).  C - V^T V + diag against NumPy; twice, the results must be bitwise equal."""
    rng = np.random.default_rng(m + k)
    V = rng.standard_normal((k, m))
    C0 = rng.standard_normal((m, m))
    ref = C0 - V.T @ V + 0.3 * np.eye(m)
    if lower:
        ref = np.tril(ref) + np.triu(C0, 1)
    dV = up(rt, V)
    outs = []
    for _ in range(2):
        dC = up(rt, C0)
        ok(rt.lib.gpx_syrk_sub(rt.h, k, m, P(dV), dV.shape[1], P(dC), dC.shape[1], 0.3, 1 if lower else 0))
        outs.append(down(rt, dC, m, m))
    assert relerr(outs[0], ref) < 1e-13
    assert np.array_equal(outs[0], outs[1])
