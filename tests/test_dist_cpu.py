"""world_size > 1 control flow of gaussian_process_amd.distributed (2-D block-cyclic, sharded) under
gloo on CPU.

The per-rank kernels are replaced by the NumPy test double (tests/dist_numpy_ops.py); what is
under test is everything the ranks do TOGETHER: tile ownership, look-ahead, the diagonal-tile / row
broadcasts and the transposed exchange on the row / column sub-communicators, the ring buffers of
the delayed updates, the info reduction, the fan-in triangular solves, prediction and the gradient
by slabs.  The same DistributedGP code runs with GpuOps + RCCL on the GPU box
(tests/test_gpu_dist.py, bench.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_worker import run as _run


def _check(outs, case, world):
    assert len(outs) == world
    for o in outs:
        assert o["shapes"] == ((case["m"],), (case["m"], case["m"]))
        assert o["L"] < 1e-12 and o["dinv"] < 1e-9, o
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-9, o
        assert o["mean"] < 1e-10 and o["cov"] < 1e-9 and o["var"] < 1e-9 and o["res"] < 1e-10, o
    # delayed updates over enough tiles: the grouped launch (several tile columns in one product) must have run
    if case.get("update_k", 2048) > case["block"] and case["n"] >= 10 * case["block"] and not case.get("grid"):
        assert max(o["grouped"] for o in outs) > 0, [o["grouped"] for o in outs]
    return outs


CASES2 = [
    dict(n=640, m=33, d=3, block=128, kind="ard"),      # 5 tiles: odd tile count, ragged m
    dict(n=700, m=64, d=8, block=256, kind="ard"),      # ragged last tile (700 = 2*256 + 188)
    dict(n=385, m=7, d=1, block=128, kind="rbf"),       # odd n: last tile is a single row
    dict(n=100, m=5, d=2, block=128, kind="ard"),       # one tile: rank 1 owns nothing
    dict(n=1500, m=9, d=2, block=128, kind="ard", update_k=384),   # delayed updates in groups of 3 panels (ring of 5)
    dict(n=1500, m=9, d=2, block=128, kind="ard", update_k=128, grad_block=384),   # no delay; gradient slabs of 3 tiles
    dict(n=1100, m=6, d=3, block=256, kind="ard", grid=(1, 2)),    # 1 x 2: the row broadcast carries the whole panel
    dict(n=900, m=6, d=3, block=128, kind="ard", force_collectives=True),   # 2 x 1 with the one-rank row groups in use
    # tile solves by the tiles' explicit inverses (many right-hand sides) + gradient slabs of 2 tiles, ragged last tile
    dict(n=1400, m=70, d=3, block=256, kind="ard", update_k=512, grad_block=512, many_rhs=64),
]


@pytest.mark.parametrize("case", CASES2, ids=lambda c: "n%d_b%d_%s_u%d_g%s" % (c["n"], c["block"], c["kind"], c.get("update_k", 2048), c.get("grid")))
def test_world2_matches_oracle(case):
    outs = _check(_run(2, case), case, 2)
    assert outs[0]["grid"] == tuple(case.get("grid", (2, 1)))


def test_world3_matches_oracle():
    case = dict(n=900, m=20, d=4, block=128, kind="ard", update_k=256)      # 1 x 3 grid
    _check(_run(3, case), case, 3)


@pytest.mark.parametrize("world,case", [
    (4, dict(n=1700, m=10, d=3, block=128, kind="ard", update_k=256, grad_block=128)),           # 2 x 2, 14 tiles
    (4, dict(n=1300, m=10, d=3, block=128, kind="ard", update_k=256, grid=(4, 1))),              # 4 x 1
    (6, dict(n=1500, m=11, d=2, block=128, kind="ard", update_k=256, grad_block=256)),           # 2 x 3: P does not divide Q
    (8, dict(n=2200, m=12, d=2, block=128, kind="ard", update_k=256, grad_block=512)),           # 2 x 4, 18 tiles
    (4, dict(n=2100, m=80, d=3, block=256, kind="ard", update_k=512, grad_block=768, many_rhs=64)),   # 2 x 2, inverse-tile solves, slabs of 3 tiles
    (8, dict(n=1800, m=12, d=2, block=128, kind="ard", update_k=384, grad_block=256, grid=(4, 2))),   # 4 x 2
], ids=["world4_2x2", "world4_4x1", "world6_2x3", "world8_2x4", "world4_2x2_tile_inverses", "world8_4x2"])
def test_more_ranks_match_oracle(world, case):
    _check(_run(world, case), case, world)


def test_resident_matrix_is_sharded():
    """Per-rank resident matrix bytes ~ lower triangle / (P Q): nothing is replicated."""
    case = dict(n=4096, m=4, d=2, block=128, kind="ard", update_k=256, grad_block=1024)
    outs = _check(_run(4, case), case, 4)
    for o in outs:
        mem = o["mem"]
        assert mem["matrix"] <= 1.15 * mem["lower_over_world"], mem          # 32 tiles: diagonal tiles' upper halves weigh ~6 %
        assert mem["matrix"] >= 0.9 * mem["lower_over_world"], mem


@pytest.mark.parametrize("world", [1, 2])
def test_panel_workspaces_stay_below_half_of_the_matrix(world):
    """with the default tile / update-K choices the rings of pending panels and the staging
    buffers of the collectives take at most half of a rank's share of the matrix on one and two ranks (round 2:
    19.3 GB beside 17.7 GB at world 1).  Same tile COUNT as the headline (n / nb = 32 resp. 64 tiles), scaled down
    8-fold: the ratio depends only on n / nb, the update K / nb and the grid."""
    from gaussian_process_amd.distributed import default_block
    scale = 8
    case = dict(n=65536 // scale, m=3, d=2, block=default_block(world) // scale, kind="ard", alloc_only=True)
    # (update_k scales with the tile: the defaults are 2 tiles per update on one and two ranks)
    outs = _run(world, dict(case, update_k={1: 4096, 2: 2048}[world] // scale))
    for o in outs:
        assert o["mem"]["panels"] <= 0.5 * o["mem"]["matrix"], o


def test_world2_not_positive_definite_reported_on_every_rank():
    # duplicated-point Gram with negative "noise": first failing minor is rank-independent
    outs = _run(2, dict(n=400, m=4, d=2, block=128, kind="ard", noise=-0.5, expect_fail=True))
    assert outs[0]["info"] > 0 and outs[0]["info"] == outs[1]["info"]
    for o in outs:
        assert o["raised"] is not None and "not positive definite" in o["raised"]


@pytest.mark.parametrize("world", [2, 4])
def test_estimator_interface_and_empirical_bayes(world):
    """DistributedGaussianProcessRegression: the reference's estimator API (regression.py:8-76,
    base.py:21-58) over the ranks -- values against the oracle, identical optimiser path on all."""
    case = dict(n=300, m=21, d=2, block=128, kind="ard", estimator=True)
    outs = _run(world, case)
    for o in outs:
        assert o["keys"] == ["sigma", "cov__sigma", "cov__beta"]
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-8, o
        assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["mean_only"] < 1e-9 and o["var"] < 1e-9, o
        assert o["eb_gain"] > 0.0 and o["score"] > 0.5, o
    assert all(o["theta"] == outs[0]["theta"] for o in outs)          # the ranks stayed in lockstep, bit for bit
