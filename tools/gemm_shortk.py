"""Duration of one short-K GEMM launch against K (development tool; run under rocprofv3 --kernel-trace and read
the per-launch durations): is a K = 128 panel solve bound by the DMA latency of each k-tile?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime
rt = runtime(0); rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
ld = 65536
A = torch.randn(4096, ld, dtype=torch.float64, device="cuda")
B = torch.randn(4096, ld, dtype=torch.float64, device="cuda")
Cm = torch.zeros(4096, ld, dtype=torch.float64, device="cuda")
for M, N in ((3968, 128), (2048, 2048)):
    for K in (16, 32, 64, 128, 256, 512, 1024):
        for rep in range(3):
            _lib.check(rt.lib.gpx_gemm(rt.h, 0, 1, M, N, K, -1.0, P(A), ld, P(B), ld, 1.0, P(Cm), ld), "gemm")
            torch.cuda.synchronize()
            print("LAUNCH %d %d %d" % (M, N, K), file=sys.stderr, flush=True)
