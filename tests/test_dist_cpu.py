"""world_size > 1 control flow of gaussian_process_amd.distributed under gloo on CPU.

The per-rank kernels are replaced by the NumPy test double (tests/dist_numpy_ops.py); what is
under test is everything the ranks do TOGETHER: block-column ownership, look-ahead, the packed
panel broadcasts, double-buffered staging, the info reduction, the RHS-split predict.  The same
DistributedGP code runs with GpuOps + RCCL on the GPU box (tests/test_gpu_dist.py, bench.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_worker import run as _run


CASES = [
    dict(n=640, m=33, d=3, block=128, kind="ard"),      # 5 blocks: odd block count, ragged m
    dict(n=700, m=64, d=8, block=256, kind="ard"),      # ragged last block (700 = 2*256 + 188)
    dict(n=385, m=7, d=1, block=128, kind="rbf"),       # odd n: last block is a single row
    dict(n=100, m=5, d=2, block=128, kind="ard"),       # one block: rank 1 owns nothing
    dict(n=1500, m=9, d=2, block=128, kind="ard", update_k=384),   # delayed updates in groups of 3 panels
    dict(n=1500, m=9, d=2, block=128, kind="ard", update_k=384, row_chunk=256),   # panels sent in row pieces
    dict(n=1100, m=6, d=3, block=256, kind="ard", row_chunk=256),  # piece = block, ragged last piece
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_b%d_%s_u%d_r%s" % (c["n"], c["block"], c["kind"], c.get("update_k", 2048), c.get("row_chunk")))
def test_world2_matches_oracle(case):
    for o in _run(2, case):
        assert o["shapes"] == ((case["m"],), (case["m"], case["m"]))
        assert o["L"] < 1e-12 and o["dinv"] < 1e-9, o
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-9, o
        assert o["mean"] < 1e-10 and o["cov"] < 1e-9 and o["res"] < 1e-10, o


def test_world3_matches_oracle():
    for o in _run(3, dict(n=900, m=20, d=4, block=128, kind="ard", update_k=256, row_chunk=384)):
        assert o["L"] < 1e-12 and o["lml"] < 1e-11 and o["mean"] < 1e-10 and o["cov"] < 1e-9, o


@pytest.mark.parametrize("world,case", [
    (4, dict(n=1700, m=10, d=3, block=128, kind="ard", update_k=256, row_chunk=256, grad_block=128)),   # 14 blocks on 4 ranks
    (8, dict(n=2200, m=12, d=2, block=128, kind="ard", update_k=256, row_chunk=512, grad_block=128)),   # 18 blocks on 8 ranks
], ids=["world4", "world8"])
def test_more_ranks_match_oracle(world, case):
    """The ownership / look-ahead / deferral pattern of the 4- and 8-rank runs (owner of k+2 is another
    rank than the owners of k and k+1; some ranks own 2 columns, some 3)."""
    outs = _run(world, case)
    assert len(outs) == world
    for o in outs:
        assert o["L"] < 1e-12 and o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-9, o
        assert o["mean"] < 1e-10 and o["cov"] < 1e-9 and o["res"] < 1e-10, o


def test_world2_not_positive_definite_reported_on_every_rank():
    # duplicated-point Gram with negative "noise": first failing minor is rank-independent
    outs = _run(2, dict(n=400, m=4, d=2, block=128, kind="ard", noise=-0.5, expect_fail=True))
    assert outs[0]["info"] > 0 and outs[0]["info"] == outs[1]["info"]
    for o in outs:
        assert o["raised"] is not None and "not positive definite" in o["raised"]


def test_world2_estimator_interface_and_empirical_bayes():
    """DistributedGaussianProcessRegression: the reference's estimator API (regression.py:8-76,
    base.py:21-58) over two ranks -- values against the oracle, identical optimiser path on both."""
    case = dict(n=300, m=21, d=2, block=128, kind="ard", estimator=True)
    outs = _run(2, case)
    for o in outs:
        assert o["keys"] == ["sigma", "cov__sigma", "cov__beta"]
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-8, o
        assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["mean_only"] < 1e-9, o
        assert o["eb_gain"] > 0.0 and o["score"] > 0.5, o
    assert outs[0]["theta"] == outs[1]["theta"]          # the ranks stayed in lockstep, bit for bit
