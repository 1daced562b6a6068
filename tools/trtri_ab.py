"""trtri + lauum timing at n (development A/B; switches come from the environment)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rt = runtime(0); rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
X = rt.upload_matrix(np.random.default_rng(0).uniform(-3, 3, (n, 8)))
K = rt.matrix(n, n); kd = rt.vector(n); dinv = rt.dinv(n)
beta, bp = _lib.host_doubles(0.5 * np.logspace(-2, 0, 8))
res = []
for rep in range(2):
    _lib.check(rt.lib.gpx_gram(rt.h, 1, P(X), n, X.shape[1], None, 0, 0, 8, 1.0, bp, 0.1, P(K), K.shape[1], 1), "gram")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(rt.lib.gpx_potrf(rt.h, n, P(K), K.shape[1], P(dinv)), "potrf")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    _lib.check(rt.lib.gpx_trtri(rt.h, n, P(K), K.shape[1], P(dinv)), "trtri")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    _lib.check(rt.lib.gpx_lauum_upper(rt.h, n, P(K), K.shape[1], P(kd)), "lauum")
    torch.cuda.synchronize(); t3 = time.perf_counter()
    res.append((t1 - t0, t2 - t1, t3 - t2))
print("LPT=%s n=%d potrf %.1f ms trtri %.1f ms lauum %.1f ms" % (os.environ.get("GPX_GEMM_LPT", "default"), n,
      1e3 * res[-1][0], 1e3 * res[-1][1], 1e3 * res[-1][2]), flush=True)
