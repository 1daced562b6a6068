"""bench.py's N > 1 leg: one rank per GPU, RCCL, strong scaling at fixed n (BASELINE configs[3]).

A step is the same fit+predict pass as at N = 1: Gram of the owned block
columns, distributed potrf (distributed.py), replicated alpha / LML, predict split over the test
columns, and the LML gradient from per-owner block columns of Ky^-1.
value = n^3/3 / (potrf time, max over ranks)."""
import json
import os
import time

import numpy as np

FP64_MFMA_PEAK_TFLOPS = 78.6


def main(args, rank, local_rank, world):
    import torch
    import torch.distributed as dist
    from . import _lib
    from .device import runtime
    from .distributed import DistributedGP, GpuOps

    backend = os.environ.get("GPX_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)          # several ranks on one GPU only in gloo rehearsals
    torch.cuda.set_device(dev_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group(backend)
    rt = runtime(dev_index)
    rt.bind_stream()
    # synthetic inputs: the generator is deterministic, so every rank builds the same data
    from .datasets import synthetic_ard
    n, m, d = args.n, args.m, 8
    X, y, Xs, beta = synthetic_ard(n, m=m, d=d, seed=0, noise=0.1)
    gp = DistributedGP(GpuOps(rt), dist, rank, world, block=getattr(args, "block", 0) or {1: 2048, 2: 2048, 4: 1024, 8: 512}.get(world, 1024),
                       row_chunk=int(os.environ.get("GPX_ROW_CHUNK", "0")) or None,   # env: development overrides
                       grad_block=int(os.environ.get("GPX_GRAD_BLOCK", "0")) or None,
                       update_k=int(os.environ.get("GPX_UPDATE_K", "0")) or None)
    sigma, noise = 1.0, 0.1

    def step(marks):
        def mark(name):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))
        mark("start")
        gp.build_gram()
        mark("gram")
        gp.potrf()
        mark("potrf")
        lml = gp.solve_lml()
        mark("solve_lml")
        mean, cov = gp.predict_device(gp_Xs)
        mark("predict")
        grad = gp.grad()
        mark("grad")
        return lml, mean, cov, grad

    # allocate once (fit() would also upload): buffers stay resident across steps
    gp.allocate(X, y, "ard", sigma, beta, noise)
    gp_Xs = gp.upload_test(Xs)
    for _ in range(args.warmup):
        step([])
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    all_marks = []
    for _ in range(args.steps):
        mk = []
        lml, mean, cov, grad = step(mk)
        all_marks.append(mk)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stages = {}
    for mk in all_marks:
        for (n0, e0), (n1, e1) in zip(mk[:-1], mk[1:]):
            stages[n1] = stages.get(n1, 0.0) + e0.elapsed_time(e1) / args.steps
    names = ["gram", "potrf", "solve_lml", "predict", "grad"]
    vec = torch.tensor([wall] + [stages[k] for k in names], dtype=torch.float64, device=rt.dev)
    dist.all_reduce(vec, op=dist.ReduceOp.MAX)
    vec = vec.cpu().numpy()
    wall = float(vec[0])
    stages = {k: float(v) for k, v in zip(names, vec[1:])}
    # parity property at full size, on every rank's replica of alpha: || Ky alpha - y || / || y ||
    res = gp.residual()
    if rank == 0:
        potrf_tflops = n ** 3 / 3 / (stages["potrf"] / 1e3) / 1e12
        ms = 1e3 * wall / args.steps
        out = {
            "metric": "GP fit+predict wall time & potrf fp64 TFLOP/s at n=%d d=8" % n,
            "value": round(potrf_tflops, 3), "unit": "TFLOP/s (potrf, n^3/3 flops)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "synthetic ARD regression n=%d d=8 m=%d, noise 0.1 (BASELINE configs[3])" % (n, m),
                       "parallelism": "block-column-cyclic x%d, block %d, panel broadcasts (%s) in pieces of %d rows, look-ahead 1"
                                      % (world, gp.block, backend, gp.row_chunk),
                       "step": "gram + potrf + alpha/LML + predict + LML gradient (block columns of Ky^-1 per owner)"},
            "fit_predict_wall_s": round(ms / 1e3, 4),
            "fit_predict_without_gradient_s": round(sum(stages[k] for k in names[:4]) / 1e3, 4),
            "stages_ms": {k: round(v, 3) for k, v in stages.items()},
            "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel (potrf phase, whole job)",
                         "achieved": round(potrf_tflops, 3), "peak": FP64_MFMA_PEAK_TFLOPS * world,
                         "unit": "TFLOP/s", "frac": round(potrf_tflops / (FP64_MFMA_PEAK_TFLOPS * world), 4),
                         "traffic": None},
            "parity": {"lml": lml, "residual_Ky_alpha_minus_y_rel": res,
                       "mean_finite": bool(np.isfinite(mean.cpu().numpy()).all()),
                       "grad": [float(g) for g in grad]},
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0
