"""Seeded random shapes through the C ABI: ragged sizes, every operand layout, multi-round launches,
single-row / single-column cases -- the cases a hand-picked list misses.  NumPy/SciPy are the checkers
(fp64, relative Frobenius <= 1e-12 unless the conditioning says otherwise)."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

from conftest import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    from gaussian_process_amd.device import runtime
    r = runtime()
    r.bind_stream()
    return r


def P(t):
    return C.c_void_p(t.data_ptr())


def ok(rc):
    from gaussian_process_amd import _lib
    _lib.check(rc, "gpx call")


def _shapes(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        kind = rng.integers(0, 4)
        if kind == 0:      # small and ragged
            out.append((int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 70))))
        elif kind == 1:    # mid-size, odd
            out.append((int(rng.integers(200, 1500)), int(rng.integers(200, 1500)), int(rng.integers(1, 300))))
        elif kind == 2:    # skinny with many row tiles (multi-round launches, possibly a separate tail)
            out.append((int(rng.integers(8000, 30000)), int(rng.integers(1, 700)), int(rng.integers(1, 50))))
        else:              # wide
            out.append((int(rng.integers(1, 700)), int(rng.integers(8000, 30000)), int(rng.integers(1, 50))))
    return out


@pytest.mark.parametrize("idx,shape", list(enumerate(_shapes(2024, 24))))
def test_gemm_random(rt, idx, shape):
    M, N, K = shape
    rng = np.random.default_rng(1000 + idx)
    ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    alpha, beta = float(rng.uniform(-2, 2)), float(rng.choice([0.0, 1.0, rng.uniform(-1, 1)]))
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    Cm = rng.standard_normal((M, N))
    dA, dB, dC = rt.upload_matrix(A), rt.upload_matrix(B), rt.upload_matrix(Cm)
    ok(rt.lib.gpx_gemm(rt.h, ta, tb, M, N, K, alpha, P(dA), dA.shape[1], P(dB), dB.shape[1], beta, P(dC), dC.shape[1]))
    ref = alpha * ((A.T if ta else A) @ (B.T if tb else B)) + beta * Cm
    got = rt.download(dC[:M, :N])
    assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))), (M, N, K, ta, tb)


@pytest.mark.parametrize("n", [int(v) for v in np.random.default_rng(7).integers(1, 3500, size=10)])
def test_cholesky_chain_random_order(rt, n):
    """potrf -> logdet -> trsm (both directions, random nrhs) -> trtri -> lauum at a random order n."""
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 2))
    A = G @ G.T / n + np.eye(n)
    dA, dinv = rt.upload_matrix(A), rt.dinv(n)
    ld = dA.shape[1]
    ok(rt.lib.gpx_potrf(rt.h, n, P(dA), ld, P(dinv)))
    Lref = sla.cholesky(A, lower=True)
    assert relerr(np.tril(rt.download(dA[:n, :n])), Lref) < 1e-13
    nrhs = int(rng.integers(1, 400))
    Bm = rng.standard_normal((n, nrhs))
    for trans in (0, 1):
        dB = rt.upload_matrix(Bm)
        ok(rt.lib.gpx_trsm(rt.h, trans, n, nrhs, P(dA), ld, P(dinv), P(dB), dB.shape[1]))
        ref = sla.solve_triangular(Lref, Bm, lower=True, trans="T" if trans else "N")
        assert relerr(rt.download(dB[:n, :nrhs]), ref) < 1e-12
    m = int(rng.integers(1, 300))
    Xr = rng.standard_normal((m, n))
    dX = rt.upload_matrix(Xr)
    ok(rt.lib.gpx_trsm_right(rt.h, m, n, P(dA), ld, P(dinv), P(dX), dX.shape[1]))
    assert relerr(rt.download(dX[:m, :n]), sla.solve_triangular(Lref, Xr.T, lower=True).T) < 1e-12
    ok(rt.lib.gpx_trtri(rt.h, n, P(dA), ld, P(dinv)))
    assert relerr(np.tril(rt.download(dA[:n, :n])), np.linalg.inv(Lref)) < 1e-11
    Kinv = rt.matrix(n, n)
    ok(rt.lib.gpx_lauum(rt.h, n, P(dA), ld, P(Kinv), Kinv.shape[1]))
    assert relerr(np.tril(rt.download(Kinv[:n, :n])), np.tril(np.linalg.inv(A))) < 1e-11


@pytest.mark.parametrize("n,m", [(int(a), int(b)) for a, b in np.random.default_rng(11).integers(1, 2500, size=(8, 2))])
def test_syrk_sub_random(rt, n, m):
    rng = np.random.default_rng(n * 7 + m)
    V = rng.standard_normal((n, m))
    Cm = rng.standard_normal((m, m))
    Cm = Cm + Cm.T
    from gaussian_process_amd import _lib
    for uplo in (_lib.GPX_FULL, _lib.GPX_LOWER):
        dV, dC = rt.upload_matrix(V), rt.upload_matrix(Cm)
        ok(rt.lib.gpx_syrk_sub(rt.h, n, m, P(dV), dV.shape[1], P(dC), dC.shape[1], 0.25, uplo))
        ref = Cm - V.T @ V + 0.25 * np.eye(m)
        got = rt.download(dC[:m, :m])
        if uplo == _lib.GPX_LOWER:
            got, ref = np.tril(got), np.tril(ref)
        assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))
