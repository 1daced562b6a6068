"""GEMM timings for the shapes of the block-column path, operands as views of one big n x n matrix
(leading dimension n) exactly as distributed.py issues them (development A/B tool; the library is
chosen by GPX_LIB_PATH)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime

rt = runtime(0)
rt.bind_stream()
lib = rt.lib
P = lambda t: C.c_void_p(t.data_ptr())
n = 65536
big = torch.rand((n, n), dtype=torch.float64, device=rt.dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
b = 512
for (j0, K) in [(49152, 2048), (45056, 2048), (40960, 2048), (32768, 2048), (24576, 2048), (45056, 512), (32768, 512)]:
    M = n - j0
    c0 = j0 - K
    A = big[j0:n, c0:j0]
    B = big[j0:j0 + b, c0:j0]
    Cm = big[j0:n, j0:j0 + b]
    t = timeit(lambda: _lib.check(lib.gpx_gemm(rt.h, 0, 1, M, b, K, -1.0, P(A), n, P(B), n, 1.0, P(Cm), n), "gemm"))
    out.append("M=%d K=%d %.3f ms %.1f TF" % (M, K, t, 2.0 * M * b * K / t / 1e9))
print(os.path.basename(_lib.LIB_PATH), " | ".join(out), flush=True)
