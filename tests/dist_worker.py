"""Shared launcher for the multi-rank tests: spawns `world` processes that run
gaussian_process_amd.distributed.DistributedGP (2-D block-cyclic, sharded) under gloo, with either the NumPy test double
(CPU, tests/test_dist_cpu.py) or the real libgpx kernels on cuda:0 (tests/test_gpu_dist.py; all
ranks share the one GPU of the box), and compares every rank's results with the oracle."""
import os
import socket
import sys
import traceback

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, case, q):
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)   # a stuck rank names its frame and dies
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch.distributed as dist
        from gaussian_process_amd.distributed import DistributedGP, GpuOps
        from oracle import gp_oracle as orc
        torch.set_num_threads(2)
        if case.get("backend") == "nccl":
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        n, m, d, block, kind = case["n"], case["m"], case["d"], case["block"], case["kind"]
        X, y, Xs, _ = orc.synthetic_ard(n, m, d, seed=case.get("seed", 3))
        sigma, noise = 1.3, case.get("noise", 0.05)
        beta = 0.5 * np.logspace(-1, 0, d) if kind == "ard" else np.array([0.7])
        if case.get("gpu"):
            from gaussian_process_amd.device import runtime
            torch.cuda.set_device(0)
            rt = runtime(0)
            rt.bind_stream()
            ops = GpuOps(rt)
        else:
            from dist_numpy_ops import NumpyOps
            ops = NumpyOps()
        host = lambda x: x.detach().cpu().numpy()
        grid = tuple(case["grid"]) if case.get("grid") else None
        gp = DistributedGP(ops, dist, rank, world, block=block, update_k=case.get("update_k", 2048),
                           grad_block=case.get("grad_block", 256), grid=grid,
                           force_collectives=bool(case.get("force_collectives")), many_rhs=case.get("many_rhs"))
        out = {"rank": rank}
        if case.get("alloc_only"):
            # footprint only: the resident slabs and the panel workspaces as fit() would allocate them
            from gaussian_process_amd.distributed import default_block
            gp = DistributedGP(ops, dist, rank, world, block=case.get("block") or default_block(world), grid=grid,
                               update_k=case.get("update_k"))
            gp.allocate(X, y, kind, sigma, beta, noise)
            out.update(mem=gp.resident_bytes, nb=gp.nb, G=gp.G, grid=(gp.P, gp.Q))
        elif case.get("estimator"):
            # the reference's estimator interface over the ranks (distributed_regression.py)
            from gaussian_process_amd import ARD, RBF
            from gaussian_process_amd.distributed_regression import DistributedGaussianProcessRegression as DGPR
            if not case.get("gpu"):
                DGPR.ops_factory = staticmethod(lambda: ops)
            DGPR.dist_options = dict(block=block, update_k=case.get("update_k", 2048), grad_block=256, grid=grid,
                                     force_collectives=bool(case.get("force_collectives")))
            cov = ARD(sigma=sigma, beta=beta.copy()) if kind == "ard" else RBF(sigma=sigma, beta=float(beta[0]))
            est = DGPR(cov, sigma=noise)
            est.fit(X, y)
            lml = est.log_marginal_likelihood()
            g = est.grad_log_marginal_likelihood()
            mean, cov_m = est.predict_with_variance(Xs)
            lml_o, _, Lo = orc.gpr_lml(kind, X, y, sigma, beta, noise)
            go, alpha_o = orc.gpr_grad(kind, X, y, sigma, beta, noise, spd_inverse=True)
            mean_o, cov_o = orc.gpr_predict(kind, X, y, Xs, sigma, beta, noise, L=Lo)
            rel = lambda a, b: float(np.linalg.norm(np.asarray(a) - b) / max(np.linalg.norm(b), 1e-300))
            gscale = max(np.abs(v).max() for v in go.values())
            out.update(lml=abs(lml - lml_o) / abs(lml_o), alpha=rel(est.alpha_(), alpha_o), mean=rel(mean, mean_o),
                       cov=rel(cov_m, cov_o), keys=list(g.keys()),
                       grad=max(float(np.abs(g[k] - go[k]).max()) for k in go) / gscale,
                       mean_only=rel(est.predict(Xs), mean_o),
                       var=rel(est.predict_with_variance(Xs, full_cov=False)[1], np.diagonal(cov_o)))
            before = lml
            est.empirical_bayes()
            out["eb_gain"] = est.log_marginal_likelihood() - before
            out["theta"] = [float(v) for v in est.get_opt_params(["sigma", "cov__sigma", "cov__beta"])]
            out["score"] = float(est.score(X, y))
        elif case.get("expect_fail"):
            try:
                gp.fit(X, y, kind, sigma, beta, noise)
                out["raised"] = None
            except np.linalg.LinAlgError as e:
                out["raised"] = str(e)
            out["info"] = gp.info_value
        else:
            lml = gp.fit(X, y, kind, sigma, beta, noise)
            mean, cov = gp.predict(Xs)
            g = gp.grad()
            go, _ = orc.gpr_grad(kind, X, y, sigma, beta, noise, spd_inverse=True)
            go = np.concatenate([go["sigma"], go["cov__sigma"], go["cov__beta"]])
            out["grad"] = float(np.abs(g - go).max() / np.abs(go).max())
            mean_d, var_d = gp.predict(Xs, full_cov=False)
            L = gp.gather_lower()
            Lo = orc.gpr_factor(kind, X, sigma, beta, noise)
            lml_o, a_o, _ = orc.gpr_lml(kind, X, y, sigma, beta, noise, L=Lo)
            mean_o, cov_o = orc.gpr_predict(kind, X, y, Xs, sigma, beta, noise, L=Lo)
            rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
            out.update(L=rel(L, Lo), lml=abs(lml - lml_o) / abs(lml_o), alpha=rel(host(gp.alpha)[:n], orc.gpr_alpha(Lo, y)),
                       mean=rel(mean, mean_o), cov=rel(cov, cov_o), shapes=(mean.shape, cov.shape),
                       var=max(rel(var_d, np.diagonal(cov_o)), rel(mean_d, mean_o)), mem=gp.resident_bytes,
                       grid=(gp.P, gp.Q), grouped=gp.grouped_launches)
            # the diagonal-block inverses every rank keeps must match the factor it keeps
            nb = 128
            worst = 0.0
            dinv_all = gp.gather_dinv()
            for i, b0 in enumerate(range(0, n, nb)):
                b1 = min(n, b0 + nb)
                blk = dinv_all[i][: b1 - b0, : b1 - b0]
                worst = max(worst, float(np.abs(blk @ Lo[b0:b1, b0:b1] - np.eye(b1 - b0)).max()))
            out["dinv"] = worst
            out["res"] = gp.residual(chunk=case.get("res_chunk", 2048))
        q.put(out)
        dist.barrier()
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
    except Exception:
        q.put({"rank": rank, "error": traceback.format_exc()})


def run(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, case, q)) for r in range(world)]
    # CPU ranks (NumPy test double): one BLAS thread each, or `world` ranks x all cores oversubscribe the host.
    # Spawned children read the variables when they import numpy; the parent's value is restored below.
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    if not case.get("gpu"):
        for k in saved:
            os.environ[k] = "1"
    try:
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    import queue
    import time
    outs, deadline = [], time.monotonic() + 150
    while len(outs) < world:
        try:
            outs.append(q.get(timeout=1.0))
            continue
        except queue.Empty:
            pass
        dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
        if dead or time.monotonic() > deadline:
            for p in procs:
                if p.is_alive():
                    p.kill()
            raise AssertionError("ranks died or timed out without a result: exit codes %s, got %d/%d results"
                                 % ([p.exitcode for p in procs], len(outs), world))
    for p in procs:
        p.join(timeout=60)
    codes = [p.exitcode for p in procs]
    assert codes == [0] * world, "rank exit codes %s" % codes
    for o in outs:
        assert "error" not in o, o.get("error")
    return sorted(outs, key=lambda o: o["rank"])


def _nccl_world1():
    """RCCL rehearsal in THIS process: backend "nccl", world size 1, collectives forced on the one-rank
    groups -- asynchronous broadcasts / all-gathers, the forwarding stream and the stream waits of the
    production branch run for real (tests/test_gpu_dist.py::test_rccl_backend_world1_matches_oracle)."""
    import queue

    class Q:
        def __init__(self):
            self.items = []

        def put(self, x):
            self.items.append(x)

    for case in (dict(n=2500, m=37, d=8, block=128, kind="ard", gpu=True, backend="nccl", force_collectives=True,
                      update_k=512, grad_block=512),
                 dict(n=1100, m=30, d=3, block=256, kind="ard", gpu=True, backend="nccl", force_collectives=True,
                      update_k=256, grad_block=256),
                 dict(n=2300, m=70, d=3, block=512, kind="ard", gpu=True, backend="nccl", force_collectives=True,
                      update_k=512, grad_block=1024, many_rhs=64),
                 dict(n=600, m=21, d=2, block=128, kind="ard", gpu=True, backend="nccl", force_collectives=True,
                      estimator=True)):
        q = Q()
        worker(0, 1, _free_port(), case, q)
        o = q.items[0]
        assert "error" not in o, o.get("error")
        print(o, flush=True)
        if case.get("estimator"):
            assert o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-8 and o["mean"] < 1e-9 and o["cov"] < 1e-9, o
            assert o["var"] < 1e-9 and o["eb_gain"] > 0.0, o
        else:
            assert o["L"] < 1e-12 and o["dinv"] < 1e-9 and o["lml"] < 1e-11 and o["alpha"] < 1e-9 and o["grad"] < 1e-9, o
            assert o["mean"] < 1e-9 and o["cov"] < 1e-9 and o["var"] < 1e-9 and o["res"] < 1e-10, o
    print("NCCL1 OK", flush=True)


if __name__ == "__main__":
    if "--nccl1" in sys.argv:
        _nccl_world1()
