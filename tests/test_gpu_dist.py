"""The multi-rank path with the REAL kernels: DistributedGP + GpuOps (libgpx through the C ABI),
2 and 3 ranks under gloo sharing cuda:0 (the box has one GPU; RCCL refuses two ranks on one
device, so the collective transport here is gloo -- the kernels, staging, look-ahead and views
are the ones bench.py --gpus N runs over RCCL).  Compared with the oracle on every rank."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_worker import run as _run

pytestmark = pytest.mark.gpu

CASES = [
    dict(n=1536, m=100, d=8, block=256, kind="ard", gpu=True),
    dict(n=1000, m=33, d=3, block=128, kind="ard", gpu=True),     # ragged last block, ragged m
    dict(n=2049, m=64, d=1, block=512, kind="rbf", gpu=True),     # odd n: last block is one row
    dict(n=100, m=5, d=2, block=128, kind="ard", gpu=True),       # one block: rank 1 owns nothing
    dict(n=3000, m=17, d=8, block=128, kind="ard", gpu=True, update_k=512),   # delayed updates, groups of 4
    dict(n=3000, m=17, d=8, block=256, kind="ard", gpu=True, update_k=512, row_chunk=512),   # panels sent in row pieces
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_b%d_%s" % (c["n"], c["block"], c["kind"]))
def test_world2_gpu_matches_oracle(case):
    for o in _run(2, case):
        assert o["shapes"] == ((case["m"],), (case["m"], case["m"]))
        assert o["L"] < 1e-12 and o["dinv"] < 1e-9, o
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-9, o
        assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["res"] < 1e-10, o


def test_world3_gpu_matches_oracle():
    for o in _run(3, dict(n=3000, m=50, d=8, block=256, kind="ard", gpu=True, update_k=512, row_chunk=768)):
        assert o["L"] < 1e-12 and o["lml"] < 1e-11 and o["mean"] < 1e-9 and o["cov"] < 1e-9, o


def test_world4_gpu_matches_oracle():
    """Four ranks sharing the GPU (gloo): the 4-rank ownership / deferral pattern on the real kernels."""
    outs = _run(4, dict(n=4000, m=40, d=8, block=256, kind="ard", gpu=True, update_k=512, row_chunk=1024,
                        grad_block=256))
    assert len(outs) == 4
    for o in outs:
        assert o["L"] < 1e-12 and o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-9, o
        assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["res"] < 1e-10, o


def test_world2_gpu_not_positive_definite():
    outs = _run(2, dict(n=700, m=4, d=2, block=128, kind="ard", noise=-0.5, expect_fail=True, gpu=True))
    assert outs[0]["info"] > 0 and outs[0]["info"] == outs[1]["info"]
    for o in outs:
        assert o["raised"] is not None and "not positive definite" in o["raised"]


def test_world2_estimator_interface_and_empirical_bayes():
    """DistributedGaussianProcessRegression: the reference's estimator API (regression.py:8-76,
    base.py:21-58) over two ranks -- values against the oracle, identical optimiser path on both."""
    case = dict(n=500, m=21, d=2, block=128, kind="ard", estimator=True, gpu=True)
    outs = _run(2, case)
    for o in outs:
        assert o["keys"] == ["sigma", "cov__sigma", "cov__beta"]
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-8, o
        assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["mean_only"] < 1e-9, o
        assert o["eb_gain"] > 0.0 and o["score"] > 0.5, o
    assert outs[0]["theta"] == outs[1]["theta"]          # the ranks stayed in lockstep, bit for bit
