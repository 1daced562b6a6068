"""W B and W^T B with a triangular 2048-tile W (the many-right-hand-side tile solves of the multi-GPU path):
time per launch, library chosen by GPX_LIB_PATH for A/B runs.  Development tool."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime
rt = runtime(0); rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
for n, m in ((2048, 4096), (2048, 2048), (1024, 8192), (2048, 16384), (4096, 4096)):
    W = torch.tril(torch.randn(n, n, dtype=torch.float64, device="cuda"))
    B = torch.randn(n, m, dtype=torch.float64, device="cuda")
    Out = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    for trans in (0, 1):
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                _lib.check(rt.lib.gpx_trmm(rt.h, trans, n, m, P(W), n, P(B), m, P(Out), m), "trmm")
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        ref = (W.T if trans else W) @ B
        err = float((Out - ref).abs().max() / ref.abs().max())
        print("trmm trans=%d n=%d m=%d: %8.1f us  %5.1f TFLOP/s on n^2 m flops  err %.1e" % (trans, n, m, dt * 1e6, n * n * m / dt / 1e12, err), flush=True)
