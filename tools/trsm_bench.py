"""Left solves with many right-hand sides (development): L^-1 B and L^-T B at n x m."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime
rt = runtime(0); rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
out = []
for n, m in ((65536, 4096), (65536, 2048), (32768, 512), (8192, 8192), (8192, 2000)):
    X = rt.upload_matrix(np.random.default_rng(0).uniform(-3, 3, (n, 8)))
    K = rt.matrix(n, n); dinv = rt.dinv(n)
    beta, bp = _lib.host_doubles(0.5 * np.logspace(-2, 0, 8))
    _lib.check(rt.lib.gpx_gram(rt.h, 1, P(X), n, X.shape[1], None, 0, 0, 8, 1.0, bp, 0.1, P(K), K.shape[1], 1), "gram")
    _lib.check(rt.lib.gpx_potrf(rt.h, n, P(K), K.shape[1], P(dinv)), "potrf")
    B = torch.rand((n, m), dtype=torch.float64, device=rt.dev)
    for trans in (0, 1):
        best = 1e9
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(rt.lib.gpx_trsm(rt.h, trans, n, m, P(K), K.shape[1], P(dinv), P(B), m), "trsm")
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        out.append("n=%d m=%d trans=%d %.2f ms %.1f TF" % (n, m, trans, best, float(n) * n * m / best / 1e9))
    del K, B
print(" | ".join(out), flush=True)
