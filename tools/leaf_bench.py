"""Time the 128x128 Cholesky leaf (potf2 + inverse) and the inverse alone."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime
from tools.gemm_bench import P, timeit
rt = runtime(0); rt.bind_stream(); lib = rt.lib
n = 128
A = torch.rand((n, n), dtype=torch.float64, device=rt.dev)
S = A @ A.T / n + torch.eye(n, dtype=torch.float64, device=rt.dev)
W = torch.empty_like(S); dinv = rt.dinv(n)
def potrf():
    W.copy_(S); _lib.check(lib.gpx_potrf(rt.h, n, P(W), n, P(dinv)), "potrf")
t_copy = timeit(lambda: (W.copy_(S), torch.cuda.synchronize()), reps=5)
t = timeit(lambda: (potrf(), torch.cuda.synchronize()), reps=5)
print("leaf potf2+inv (incl. sync overhead): %.1f us (copy+sync alone %.1f us)" % (t * 1e3, t_copy * 1e3))
t2 = timeit(lambda: (_lib.check(lib.gpx_diag_inv(rt.h, n, P(W), n, P(dinv)), "dinv"), torch.cuda.synchronize()), reps=5)
print("trti2 only: %.1f us" % (t2 * 1e3))
