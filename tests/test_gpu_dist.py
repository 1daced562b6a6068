"""The multi-rank path with the REAL kernels: DistributedGP (2-D block-cyclic, sharded) + GpuOps (libgpx
through the C ABI).  2, 3, 4 and 8 ranks under gloo sharing cuda:0 (the box has one GPU; RCCL refuses
two ranks on one device, so the collective transport there is gloo, staged through the host -- the
kernels, slabs, ring buffers, look-ahead and views are the ones bench.py --gpus N runs over RCCL), and
the RCCL branch itself (asynchronous broadcasts / all-gathers, forwarding stream, stream waits) with
world size 1 in a process of its own.  Compared with the oracle on every rank."""
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_worker import run as _run

pytestmark = pytest.mark.gpu

CASES = [
    dict(n=1536, m=100, d=8, block=256, kind="ard", gpu=True),
    dict(n=1000, m=33, d=3, block=128, kind="ard", gpu=True),     # ragged last tile, ragged m
    dict(n=2049, m=64, d=1, block=512, kind="rbf", gpu=True),     # odd n: last tile is one row
    dict(n=100, m=5, d=2, block=128, kind="ard", gpu=True),       # one tile: rank 1 owns nothing
    dict(n=3000, m=17, d=8, block=128, kind="ard", gpu=True, update_k=512, grad_block=512),   # delayed updates, groups of 4
    dict(n=3000, m=17, d=8, block=256, kind="ard", gpu=True, update_k=256, grid=(1, 2)),      # 1 x 2 grid
    # tile solves by the tiles' explicit inverses (gpx_trtri + gpx_trmm), gradient slabs of 2 tiles, ragged tiles
    dict(n=2900, m=90, d=8, block=512, kind="ard", gpu=True, update_k=512, grad_block=1024, many_rhs=64),
]


def _check(outs, case):
    for o in outs:
        assert o["shapes"] == ((case["m"],), (case["m"], case["m"]))
        assert o["L"] < 1e-12 and o["dinv"] < 1e-9, o
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-9, o
        assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["var"] < 1e-9 and o["res"] < 1e-10, o


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_b%d_%s_g%s" % (c["n"], c["block"], c["kind"], c.get("grid")))
def test_world2_gpu_matches_oracle(case):
    _check(_run(2, case), case)


def test_world3_gpu_matches_oracle():
    case = dict(n=3000, m=50, d=8, block=256, kind="ard", gpu=True, update_k=512)
    _check(_run(3, case), case)


def test_world4_gpu_matches_oracle():
    """Four ranks sharing the GPU (gloo), 2 x 2 grid: row AND column sub-communicators on the real kernels."""
    case = dict(n=4000, m=40, d=8, block=256, kind="ard", gpu=True, update_k=512, grad_block=1024)
    outs = _run(4, case)
    assert len(outs) == 4
    _check(outs, case)
    for o in outs:
        assert o["mem"]["matrix"] <= 1.2 * o["mem"]["lower_over_world"], o["mem"]


def test_world8_gpu_matches_oracle():
    """Eight ranks sharing the GPU (gloo), the 2 x 4 grid of the 8-GPU node."""
    case = dict(n=3500, m=24, d=8, block=128, kind="ard", gpu=True, update_k=256, grad_block=512)
    outs = _run(8, case)
    assert len(outs) == 8 and outs[0]["grid"] == (2, 4)
    _check(outs, case)


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
        assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["mean_only"] < 1e-9 and o["var"] < 1e-9, o
        assert o["eb_gain"] > 0.0 and o["score"] > 0.5, o
    assert outs[0]["theta"] == outs[1]["theta"]          # the ranks stayed in lockstep, bit for bit


def test_rccl_backend_world1_matches_oracle():
    """The production backend: init_process_group("nccl") with world size 1 on cuda:0, small tiles so
    that there are many panels -- DistributedGP.fit / predict / grad and the estimator class against
    the oracle.  Runs in a child process (a process group must not leak into this one)."""
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-u", os.path.join(here, "dist_worker.py"), "--nccl1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "NCCL1 OK" in r.stdout, r.stdout[-3000:]
