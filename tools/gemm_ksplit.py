"""Does a very long K hurt the big launches?  D = C - A B^T at M = N = size with K = size in one launch
against the same product as consecutive launches of K / parts each (development tool)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime

size = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rt = runtime(0)
rt.bind_stream()
P = lambda t, off=0: C.c_void_p(t.data_ptr() + 8 * off)
a = torch.rand((size, size), dtype=torch.float64, device=rt.dev) - 0.5
b = torch.rand((size, size), dtype=torch.float64, device=rt.dev) - 0.5
c = torch.zeros((size, size), dtype=torch.float64, device=rt.dev)
for parts in (1, 2, 4, 1, 2):
    kk = size // parts
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for p in range(parts):
        _lib.check(rt.lib.gpx_gemm(rt.h, 0, 1, size, size, kk, -1.0, P(a, p * kk), size, P(b, p * kk), size, 1.0,
                                   P(c), size), "gemm")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("M=N=K=%d in %d launch(es) of K=%d: %.3f ms %.2f TFLOP/s" % (size, parts, kk, ms, 2.0 * size ** 3 / ms / 1e9), flush=True)
