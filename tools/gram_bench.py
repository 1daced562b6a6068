"""Gram-kernel timing at full size (development tool): lower-triangle Ky build at n, d = 8."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaussian_process_amd import _lib
from gaussian_process_amd.device import runtime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d = 8
rt = runtime(0)
rt.bind_stream()
P = lambda t: C.c_void_p(t.data_ptr())
X = rt.upload_matrix(np.random.default_rng(0).uniform(-3, 3, (n, d)))
K = rt.matrix(n, n)
beta, bp = _lib.host_doubles(0.5 * np.logspace(-2, 0, d))
for uplo, name, nbytes in ((1, "lower", 4.0 * n * (n + 1)), (0, "full", 8.0 * n * n)):
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(rt.lib.gpx_gram(rt.h, 1, P(X), n, X.shape[1], None, 0, 0, d, 1.0, bp, 0.1, P(K), K.shape[1], uplo), "gram")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    print("gram %s n=%d: %.3f ms  %.0f GB/s written" % (name, n, ms, nbytes / ms / 1e6), flush=True)
# how the time moves with the input dimension (VALU work per element: 2 d + ~24 instructions)
for dd in (1, 2, 4, 8):
    Xd = rt.upload_matrix(np.random.default_rng(1).uniform(-3, 3, (n, dd)))
    bd, bdp = _lib.host_doubles(0.5 * np.logspace(-2, 0, dd))
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(rt.lib.gpx_gram(rt.h, 1, P(Xd), n, Xd.shape[1], None, 0, 0, dd, 1.0, bdp, 0.1, P(K), K.shape[1], 1), "gram")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    print("gram lower d=%d: %.3f ms  %.0f GB/s" % (dd, ms, 4.0 * n * (n + 1) / ms / 1e6), flush=True)
# reference points for the write roofline: device fill (write-only) and copy (read + write)
K2 = None
for name, fn, nbytes in (("fill_", lambda: K.fill_(1.0), 8.0 * K.numel()), ("zero_", lambda: K.zero_(), 8.0 * K.numel())):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    print("%s %d B: %.3f ms  %.0f GB/s" % (name, nbytes, ms, nbytes / ms / 1e6), flush=True)
