"""Vector triangular solve timing and residual at full size (development tool)."""
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
ld = ((n + 31) // 32) * 32
L = torch.rand((n, ld), dtype=torch.float64, device=rt.dev)
L.sub_(0.5).mul_(2.0 / n ** 0.5)
for i in range(0, n, 4096):                       # (torch.tril cannot launch on a 65536^2 matrix)
    j = min(n, i + 4096)
    L[i:j, j:] = 0.0
    L[i:j, i:j] = torch.tril(L[i:j, i:j])
idx = torch.arange(n, device=rt.dev)
L[idx, idx] = 1.0 + torch.rand(n, dtype=torch.float64, device=rt.dev)
dinv = rt.dinv(n)
_lib.check(lib.gpx_diag_inv(rt.h, n, P(L), ld, P(dinv)), "diag_inv")
b = torch.rand((n, 2), dtype=torch.float64, device=rt.dev)
for trans in (0, 1):
    for rep in range(3):
        x = b.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.gpx_trsm(rt.h, trans, n, 1, P(L), ld, P(dinv), P(x), 2), "trsv")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    y = rt.vector(n)
    xc = x[:, 0].contiguous()
    _lib.check(lib.gpx_gemv(rt.h, 1 if trans else 0, n, n, P(L), ld, P(xc), P(y)), "gemv")
    r = y[:n] - b[:, 0]
    print("trsv trans=%d n=%d: %.3f ms  (%.0f GB/s of L's triangle)  residual %.2e" %
          (trans, n, ms, 4.0 * n * n / ms / 1e6, (r.norm() / b[:, 0].norm()).item()), flush=True)
# the streaming counterpart once W = L^-1 is there: y = W x, y = W^T x (timed on the same triangle)
xv, yv = rt.vector(n), rt.vector(n)
xv[:n] = b[:, 0]
for trans in (0, 1):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.gpx_trmv(rt.h, trans, n, P(L), ld, P(xv), P(yv)), "trmv")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    ref = rt.vector(n)
    _lib.check(lib.gpx_gemv(rt.h, 1 if trans else 0, n, n, P(L), ld, P(xv), P(ref)), "gemv")   # L is stored with zeros above
    print("trmv trans=%d n=%d: %.3f ms  (%.0f GB/s of W's triangle)  difference to the full product %.2e" %
          (trans, n, ms, 4.0 * n * n / ms / 1e6, ((yv[:n] - ref[:n]).norm() / ref[:n].norm()).item()), flush=True)
