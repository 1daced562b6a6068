"""Does the leading dimension (row stride) change GEMM throughput?  (TLB reach probe)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime
from tools.gemm_bench import P, timeit

rt = runtime(0); rt.bind_stream(); lib = rt.lib
n = 16384
for ld in (16384, 16384 + 32, 32768, 65536, 65536 + 32):
    A = torch.rand((n, ld), dtype=torch.float64, device=rt.dev) - 0.5
    B = torch.rand((n, ld), dtype=torch.float64, device=rt.dev) - 0.5
    Cm = torch.zeros((n, ld), dtype=torch.float64, device=rt.dev)
    for ta, tb, name in ((0, 1, "NT"), (0, 0, "NN"), (1, 0, "TN")):
        ms = timeit(lambda: _lib.check(lib.gpx_gemm(rt.h, ta, tb, n, n, n, 1.0, P(A), ld, P(B), ld, 0.0, P(Cm), ld), "gemm"), reps=2)
        print("ld=%d %s: %.2f ms %.2f TFLOP/s" % (ld, name, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
    del A, B, Cm
