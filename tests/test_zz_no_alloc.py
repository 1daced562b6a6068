"""gpx_gemv_static: the allocation-free variant of gpx_gemv (partials in the library's file-scope
device array) against gpx_gemv itself, through the C ABI, two calls in a row, the second smaller.

Collected last on purpose (file name): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(rt, n, m, trans, seed):
    import torch
    from gaussian_process_amd.device import ptr
    lib = rt.lib
    rng = np.random.default_rng(seed)
    A_h = rng.standard_normal((n, m))
    x_h = rng.standard_normal(n if trans else m)
    A = rt.upload_matrix(A_h)
    x = rt.upload_vector(x_h)
    nout = m if trans else n
    y_ref = rt.vector(nout, zero=True)
    y_new = rt.vector(nout, zero=True)
    rc = lib.gpx_gemv(rt.h, trans, n, m, ptr(A), A.shape[1], ptr(x), ptr(y_ref))
    assert rc == 0, rc
    if trans:
        # poison the shared array: the same call on a NaN vector leaves NaN in every partial-sum slot this
        # shape uses; the real call below has to overwrite them all
        xnan = rt.upload_vector(np.full(n, np.nan))
        ytmp = rt.vector(nout, zero=True)
        rc = lib.gpx_gemv_static(rt.h, trans, n, m, ptr(A), A.shape[1], ptr(xnan), ptr(ytmp))
        assert rc == 0, (rc, lib.gpx_last_error())
        torch.cuda.synchronize()
        assert np.isnan(rt.download(ytmp[:nout])).all()
    # the property the entry point exists for: the call allocates and frees nothing on the device
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(rt.dev)[0]
    rc = lib.gpx_gemv_static(rt.h, trans, n, m, ptr(A), A.shape[1], ptr(x), ptr(y_new))
    assert rc == 0, (rc, lib.gpx_last_error())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(rt.dev)[0] == free0
    a = rt.download(y_ref[:nout])
    b = rt.download(y_new[:nout])
    exact = A_h.T @ x_h if trans else A_h @ x_h
    scale = np.abs(exact).max()
    # fp64 sums; the two entry points may chunk the rows differently (scratch sizes differ): 1e-13 of the scale
    assert np.abs(a - exact).max() <= 1e-12 * scale
    assert np.abs(b - a).max() <= 1e-13 * scale, float(np.abs(b - a).max() / scale)
    return b


def test_gemv_static_matches_gemv_two_calls_in_a_row():
    from gaussian_process_amd.device import runtime
    rt = runtime(0)
    rt.bind_stream()
    # transposed product (the path that uses the scratch): a large call, then a smaller one that
    # leaves most of the previous call's partials in the shared array
    _run(rt, 20000, 3000, 1, seed=1)
    _run(rt, 777, 130, 1, seed=2)
    _run(rt, 5, 3, 1, seed=3)
    # many columns: the chunk height is doubled until the partials fit the 2 MiB array
    _run(rt, 4096, 70000, 1, seed=4)
    _run(rt, 300, 41, 0, seed=5)


def test_gemv_static_rejects_sizes_beyond_the_scratch():
    from gaussian_process_amd.device import ptr, runtime
    rt = runtime(0)
    rt.bind_stream()
    A = rt.matrix(2, 64)
    x = rt.vector(2)
    y = rt.vector(2)
    m_big = (1 << 18) + 2
    rc = rt.lib.gpx_gemv_static(rt.h, 1, 2, m_big, ptr(A), m_big, ptr(x), ptr(y))
    assert rc == -1
    assert b"exceeds the static scratch" in rt.lib.gpx_last_error()
    # inside the old documented box (m <= 2^18, n <= 2^28) but two chunks of 2^18 partials: rejected up front with
    # GPX_EINVAL, not GPX_ENOMEM from inside the launch (nothing is read: the check comes before any access)
    rc = rt.lib.gpx_gemv_static(rt.h, 1, (1 << 20) + 1, 1 << 18, ptr(A), 1 << 18, ptr(x), ptr(y))
    assert rc == -1
    assert b"exceeds the static scratch" in rt.lib.gpx_last_error()
    rc = rt.lib.gpx_gemv_static(rt.h, 1, (1 << 21) + 5, (1 << 17) + 2, ptr(A), (1 << 17) + 2, ptr(x), ptr(y))
    assert rc == -1
    assert b"exceeds the static scratch" in rt.lib.gpx_last_error()
