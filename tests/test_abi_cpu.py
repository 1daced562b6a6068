"""CPU-only checks of the drop-in boundary: libgpx.so loads and exports every symbol that
include/gpx.h declares; the Python mirror keeps the reference's parameter protocol."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

from gaussian_process_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libgpx.so does not export %s" % n
        assert n in _lib.SIGNATURES, "no ctypes prototype for %s" % n
    assert set(_lib.SIGNATURES) == set(names)
    assert b"gfx950" in lib.gpx_version()


def test_sizes_without_gpu():
    lib = _lib.load()
    assert lib.gpx_dinv_elems(1) == 128 * 128
    assert lib.gpx_dinv_elems(129) == 2 * 128 * 128
    assert lib.gpx_dinv_elems(65536) == 512 * 128 * 128


def test_parameter_protocol_matches_reference():
    """get_params / set_params names and bounds (regression.py:11-16, base.py:21-46)."""
    from gaussian_process_amd import GaussianProcessRegression, RBF, ARD
    gp = GaussianProcessRegression(cov=ARD(beta=[1, 1]))
    p = gp.get_params()
    assert sorted(p) == ["cov", "cov__beta", "cov__sigma", "sigma"]
    assert gp.sigma == 0.01
    assert gp.param_bounds == {"sigma": (1e-5, None), "cov__sigma": (1e-5, None),
                               "cov__beta": [(1e-5, None), (1e-5, None)]}
    names = [k for k, v in p.items() if np.isscalar(v) or isinstance(v, np.ndarray)]
    assert names == ["cov__beta", "cov__sigma", "sigma"]
    x = gp.get_opt_params(names)
    assert x.tolist() == [1.0, 1.0, 1.0, 0.01]
    gp.set_opt_params(np.array([2.0, 3.0, 4.0, 0.5]), names)
    assert gp.cov.beta.tolist() == [2.0, 3.0] and gp.cov.sigma == 4.0 and gp.sigma == 0.5
    assert gp._opt_bounds(names) == [(1e-5, None)] * 4
    g2 = GaussianProcessRegression(cov=RBF(), sigma=0.1)
    assert g2.param_bounds == {"sigma": (1e-5, None), "cov__beta": (1e-5, None),
                               "cov__sigma": (1e-5, None)}
    g2.set_params(cov__beta=0.3, sigma=0.2)
    assert g2.cov.beta == 0.3 and g2.sigma == 0.2


def test_diag_branch_and_asserts_need_no_gpu():
    from gaussian_process_amd import RBF, ARD
    kv = RBF(sigma=2.0)(np.zeros((5, 2)), np.zeros((3, 2)), diag=True)
    assert kv.K.tolist() == [2.0] * 3 and kv.dtheta["beta"].shape == (3,)
    kv = ARD(sigma=1.5, beta=[1, 2])(np.zeros((4, 2)), diag=True)
    assert kv.K.tolist() == [1.5] * 4 and kv.dtheta["beta"].shape == (2, 4)
    with pytest.raises(AssertionError):
        RBF()(np.zeros(3))
    with pytest.raises(AssertionError):
        ARD(beta=[1, 2])(np.zeros((3, 3)))


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gaussian_process_amd import RBF
    with pytest.raises(_lib.GpxError):
        RBF()(np.zeros((3, 1)))


def test_workload_generator_matches_the_oracles():
    """bench.py draws its inputs from the package (the product path must not import the oracle);
    the oracle keeps its own copy for the tests: both must produce the same workload."""
    import numpy as np
    from gaussian_process_amd.datasets import synthetic_ard
    from oracle import gp_oracle as O
    for a, b in zip(synthetic_ard(300, m=17, d=8, seed=3), O.synthetic_ard(300, m=17, d=8, seed=3)):
        assert np.array_equal(a, b)


def test_product_code_never_imports_the_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussian_process_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
