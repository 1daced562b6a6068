"""world_size > 1 control flow of gaussian_process_amd.distributed under gloo on CPU.

The per-rank kernels are replaced by the NumPy test double (tests/dist_numpy_ops.py); what is
under test is everything the ranks do TOGETHER: block-column ownership, look-ahead, the packed
panel broadcasts, double-buffered staging, the info reduction, the RHS-split predict.  The same
DistributedGP code runs with GpuOps + RCCL on the GPU box (tests/test_gpu_dist.py, bench.py)."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        from dist_numpy_ops import NumpyOps
        from gaussian_process_amd.distributed import DistributedGP
        from oracle import gp_oracle as orc
        torch.set_num_threads(2)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        n, m, d, block, kind = case["n"], case["m"], case["d"], case["block"], case["kind"]
        X, y, Xs, _ = orc.synthetic_ard(n, m, d, seed=case.get("seed", 3))
        sigma, noise = 1.3, case.get("noise", 0.05)
        beta = 0.5 * np.logspace(-1, 0, d) if kind == "ard" else np.array([0.7])
        gp = DistributedGP(NumpyOps(), dist, rank, world, block=block)
        out = {"rank": rank}
        if case.get("expect_fail"):
            try:
                gp.fit(X, y, kind, sigma, beta, noise)
                out["raised"] = None
            except np.linalg.LinAlgError as e:
                out["raised"] = str(e)
            out["info"] = gp.info_value
        else:
            lml = gp.fit(X, y, kind, sigma, beta, noise)
            mean, cov = gp.predict(Xs)
            L = np.tril(gp.A[:n, :n].numpy())
            Lo = orc.gpr_factor(kind, X, sigma, beta, noise)
            lml_o, a_o, _ = orc.gpr_lml(kind, X, y, sigma, beta, noise, L=Lo)
            mean_o, cov_o = orc.gpr_predict(kind, X, y, Xs, sigma, beta, noise, L=Lo)
            rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
            out.update(L=rel(L, Lo), lml=abs(lml - lml_o) / abs(lml_o), alpha=rel(gp.alpha.numpy()[:n], orc.gpr_alpha(Lo, y)),
                       mean=rel(mean, mean_o), cov=rel(cov, cov_o), shapes=(mean.shape, cov.shape))
            # the diagonal-block inverses every rank keeps must match the factor it keeps
            nb = 128
            worst = 0.0
            for i, b0 in enumerate(range(0, n, nb)):
                b1 = min(n, b0 + nb)
                blk = gp.dinv[i * nb * nb:(i + 1) * nb * nb].numpy().reshape(nb, nb)[: b1 - b0, : b1 - b0]
                worst = max(worst, float(np.abs(blk @ Lo[b0:b1, b0:b1] - np.eye(b1 - b0)).max()))
            out["dinv"] = worst
        dist.barrier()
        dist.destroy_process_group()
        q.put(out)
    except Exception:
        q.put({"rank": rank, "error": traceback.format_exc()})


def _run(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for o in outs:
        assert "error" not in o, o.get("error")
    return sorted(outs, key=lambda o: o["rank"])


CASES = [
    dict(n=640, m=33, d=3, block=128, kind="ard"),      # 5 blocks: odd block count, ragged m
    dict(n=700, m=64, d=8, block=256, kind="ard"),      # ragged last block (700 = 2*256 + 188)
    dict(n=385, m=7, d=1, block=128, kind="rbf"),       # odd n: last block is a single row
    dict(n=100, m=5, d=2, block=128, kind="ard"),       # one block: rank 1 owns nothing
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_b%d_%s" % (c["n"], c["block"], c["kind"]))
def test_world2_matches_oracle(case):
    for o in _run(2, case):
        assert o["shapes"] == ((case["m"],), (case["m"], case["m"]))
        assert o["L"] < 1e-12 and o["dinv"] < 1e-9, o
        assert o["lml"] < 1e-11 and o["alpha"] < 1e-9, o
        assert o["mean"] < 1e-10 and o["cov"] < 1e-9, o


def test_world3_matches_oracle():
    for o in _run(3, dict(n=900, m=20, d=4, block=128, kind="ard")):
        assert o["L"] < 1e-12 and o["lml"] < 1e-11 and o["mean"] < 1e-10 and o["cov"] < 1e-9, o


def test_world2_not_positive_definite_reported_on_every_rank():
    # duplicated-point Gram with negative "noise": first failing minor is rank-independent
    outs = _run(2, dict(n=400, m=4, d=2, block=128, kind="ard", noise=-0.5, expect_fail=True))
    assert outs[0]["info"] > 0 and outs[0]["info"] == outs[1]["info"]
    for o in outs:
        assert o["raised"] is not None and "not positive definite" in o["raised"]
