"""Timing of the pieces on the multi-GPU critical chain (development tool): diagonal-block potrf,
panel TRSM, one block-column update, panel pack copy, for block widths 512/1024/2048 at n rows."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rt = runtime(0)
rt.bind_stream()
lib = rt.lib
P = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, reps=3):
    best = 1e9
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for b in (512, 1024, 2048):
    G = torch.rand((b, b), dtype=torch.float64, device=rt.dev) - 0.5
    S = G @ G.T / b + torch.eye(b, dtype=torch.float64, device=rt.dev)
    W = torch.empty_like(S)
    dinv = rt.dinv(b)
    info = torch.zeros(4, dtype=torch.int32, device=rt.dev)

    def potrf():
        W.copy_(S)
        _lib.check(lib.gpx_potrf_async(rt.h, b, P(W), b, P(dinv), P(info), 0), "potrf")
    t_potrf = timeit(potrf) - timeit(lambda: W.copy_(S))
    X = torch.rand((n, b), dtype=torch.float64, device=rt.dev)
    t_trsm = timeit(lambda: _lib.check(lib.gpx_trsm_right(rt.h, n, b, P(W), b, P(dinv), P(X), b), "trsm"), reps=1)
    X = torch.rand((n, b), dtype=torch.float64, device=rt.dev)
    Cm = torch.zeros((n, b), dtype=torch.float64, device=rt.dev)
    t_upd = timeit(lambda: _lib.check(lib.gpx_gemm(rt.h, 0, 1, n, b, b, -1.0, P(X), b, P(X), b, 1.0, P(Cm), b), "gemm"))
    Y = torch.empty_like(X)
    t_copy = timeit(lambda: Y.copy_(X))
    print("b=%4d: potrf %.3f ms (%.1f TF)  trsm_right(%d rows) %.3f ms (%.1f TF)  update gemm %.3f ms (%.1f TF)  copy %.3f ms"
          % (b, t_potrf, b ** 3 / 3 / t_potrf / 1e9, n, t_trsm, n * b * b / t_trsm / 1e9, t_upd,
             2.0 * n * b * b / t_upd / 1e9, t_copy), flush=True)
