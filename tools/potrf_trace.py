"""potrf only at n (default 65536) on random SPD-ish data, once: for joining GPX_GEMM_LOG lines with a
rocprofv3 kernel trace (development tool)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rt = runtime(0); rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
X = rt.upload_matrix(np.random.default_rng(0).uniform(-3, 3, (n, 8)))
K = rt.matrix(n, n); dinv = rt.dinv(n)
beta, bp = _lib.host_doubles(0.5 * np.logspace(-2, 0, 8))
_lib.check(rt.lib.gpx_gram(rt.h, 1, P(X), n, X.shape[1], None, 0, 0, 8, 1.0, bp, 0.1, P(K), K.shape[1], 1), "gram")
torch.cuda.synchronize()
print("POTRF-BEGIN", file=sys.stderr, flush=True)
_lib.check(rt.lib.gpx_potrf(rt.h, n, P(K), K.shape[1], P(dinv)), "potrf")
torch.cuda.synchronize()
if len(sys.argv) > 2:
    kd = rt.vector(n)
    print("PHASE trtri", file=sys.stderr, flush=True)
    _lib.check(rt.lib.gpx_trtri(rt.h, n, P(K), K.shape[1], P(dinv)), "trtri")
    torch.cuda.synchronize()
    print("PHASE lauum", file=sys.stderr, flush=True)
    _lib.check(rt.lib.gpx_lauum_upper(rt.h, n, P(K), K.shape[1], P(kd)), "lauum")
    torch.cuda.synchronize()
