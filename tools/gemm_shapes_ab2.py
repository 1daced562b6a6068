"""Lone-workgroup-per-CU vs two-per-CU GEMM timing, compact vs strided operands (development A/B)."""
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
b, K = 512, 2048
for M in (2048, 4096, 8192, 12288, 16384, 17408, 18432, 20480, 24576, 28672, 32768, 36864):
    for strided in (0,):
        if strided:
            A, B, Cm, ld = big[8192:8192 + M, 0:K], big[0:b, 0:K], big[8192:8192 + M, 4096:4096 + b], n
        else:
            A = torch.rand((M, K), dtype=torch.float64, device=rt.dev)
            B = torch.rand((b, K), dtype=torch.float64, device=rt.dev)
            Cm = torch.zeros((M, b), dtype=torch.float64, device=rt.dev)
            ld = None
        t = timeit(lambda: _lib.check(lib.gpx_gemm(rt.h, 0, 1, M, b, K, -1.0, P(A), ld or K, P(B), ld or K, 1.0, P(Cm), ld or b), "gemm"))
        out.append("M=%d %s %.3f ms" % (M, "ld65536" if strided else "compact", t))
print(os.path.basename(_lib.LIB_PATH), " | ".join(out), flush=True)
