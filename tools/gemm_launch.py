"""The dominant-kernel launch of bench.py's roofline (D = C - A B^T, M = N = K = 16384) on its own,
for rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate runs)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime

size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rt = runtime(0)
rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
a = torch.rand((size, size), dtype=torch.float64, device=rt.dev) - 0.5
b = torch.rand((size, size), dtype=torch.float64, device=rt.dev) - 0.5
c = torch.zeros((size, size), dtype=torch.float64, device=rt.dev)
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(rt.lib.gpx_gemm(rt.h, 0, 1, size, size, size, -1.0, P(a), size, P(b), size, 1.0, P(c), size), "gemm")
    e1.record()
    torch.cuda.synchronize()
    print("gemm %d: %.3f ms %.2f TFLOP/s" % (size, e0.elapsed_time(e1), 2.0 * size ** 3 / e0.elapsed_time(e1) / 1e9), flush=True)
