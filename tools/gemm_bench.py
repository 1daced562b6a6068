"""Kernel-level timing of libgpx pieces on one MI355X (development tool, not the headline bench)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime


def P(t):
    return C.c_void_p(t.data_ptr())


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


def main():
    rt = runtime(0)
    rt.bind_stream()
    lib = rt.lib
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]
    for n in sizes:
        A = torch.rand((n, n), dtype=torch.float64, device=rt.dev) - 0.5
        B = torch.rand((n, n), dtype=torch.float64, device=rt.dev) - 0.5
        Cm = torch.zeros((n, n), dtype=torch.float64, device=rt.dev)
        for ta, tb, name in ((0, 1, "NT(KC,KC)"), (0, 0, "NN(KC,RC)"), (1, 0, "TN(RC,RC)")):
            ms = timeit(lambda: _lib.check(lib.gpx_gemm(rt.h, ta, tb, n, n, n, 1.0, P(A), n, P(B), n, 0.0,
                                                        P(Cm), n), "gemm"))
            print("gemm %s n=%d: %.2f ms  %.2f TFLOP/s" % (name, n, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
        # potrf on an SPD matrix
        Sp = (A @ A.T) / n + torch.eye(n, dtype=torch.float64, device=rt.dev)
        W = torch.empty_like(Sp)
        dinv = rt.dinv(n)

        def potrf():
            W.copy_(Sp)
            _lib.check(lib.gpx_potrf(rt.h, n, P(W), n, P(dinv)), "potrf")
        ms_copy = timeit(lambda: W.copy_(Sp))
        ms = timeit(potrf) - ms_copy
        print("potrf n=%d: %.2f ms  %.2f TFLOP/s" % (n, ms, n ** 3 / 3 / ms / 1e9), flush=True)
        rhs = torch.rand((n, 2), dtype=torch.float64, device=rt.dev)
        ms = timeit(lambda: _lib.check(lib.gpx_trsm(rt.h, 0, n, 1, P(W), n, P(dinv), P(rhs), 2), "trsv"))
        print("trsv n=%d: %.3f ms  (%.1f GB/s of L)" % (n, ms, 4.0 * n * n / ms / 1e6), flush=True)
        del A, B, Cm, Sp, W


if __name__ == "__main__":
    main()
