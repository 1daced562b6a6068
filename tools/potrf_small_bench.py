"""potrf time at small / mid sizes (development tool): how the panel-loop threshold (env
GPX_POTRF_PANELS, read once per process) moves the diagonal-block cost."""
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
sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192, 16384]
out = []
for n in sizes:
    G = torch.rand((n, 64), dtype=torch.float64, device=rt.dev) - 0.5
    S = G @ G.T / 64 + torch.eye(n, dtype=torch.float64, device=rt.dev)
    W = torch.empty_like(S)
    dinv = rt.dinv(n)
    info = torch.zeros(4, dtype=torch.int32, device=rt.dev)
    best = 1e9
    for rep in range(4):
        W.copy_(S)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.gpx_potrf_async(rt.h, n, P(W), n, P(dinv), P(info), 0), "potrf")
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    if n <= 16384:
        L = torch.tril(W)
        err = float((L @ L.T - S).abs().max())
    else:                                # (torch.tril cannot launch on a larger matrix): sampled entries of L L^T - S
        err = 0.0
        for i in range(n - 1, 0, -max(1, n // 48)):
            j = i // 2
            err = max(err, abs(float(W[i, : i + 1] @ W[i, : i + 1] - S[i, i])), abs(float(W[i, : j + 1] @ W[j, : j + 1] - S[i, j])))
    out.append("n=%d %.3f ms %.1f TF err %.1e" % (n, best, n ** 3 / 3 / best / 1e9, err))
print("PANELS=%s: " % os.environ.get("GPX_POTRF_PANELS", "default") + " | ".join(out), flush=True)
