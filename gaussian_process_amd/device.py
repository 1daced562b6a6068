"""Device-side plumbing: one gpx handle per GPU, torch tensors as the owners of HBM buffers.

PyTorch is used only for allocation, H2D/D2H copies and stream identity; all arithmetic is in
libgpx.so."""
import ctypes as C
import threading

import numpy as np

from . import _lib

_lock = threading.Lock()
_runtimes = {}


def _round_up(x, m):
    return ((x + m - 1) // m) * m


class Runtime:
    """Per-device state: gpx handle bound to torch's current stream."""

    def __init__(self, index):
        import torch
        if not torch.cuda.is_available():
            raise _lib.GpxError("no GPU visible: gaussian_process_amd needs an MI355X (no CPU fallback)")
        self.torch = torch
        self.index = index
        self.dev = torch.device("cuda", index)
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.gpx_create(C.byref(h), index), "gpx_create")
        self.h = h
        # A gpx handle is single-threaded and single-stream (one reduction scratch, one pinned staging
        # buffer, one stream binding): every estimator operation holds this lock from bind_stream() to its
        # last fetch (see on_device below).
        self.lock = threading.RLock()
        self._stream = None

    def bind_stream(self):
        """Bind the handle to torch's current stream.  When the stream changes, the new one first waits
        for everything the old one was given: the handle's scratch buffers are shared, so work on two
        streams must not overlap."""
        cur = self.torch.cuda.current_stream(self.dev)
        old = self._stream
        if old is not None and old.cuda_stream != cur.cuda_stream:
            ev = self.torch.cuda.Event()
            ev.record(old)
            cur.wait_event(ev)
        self._stream = cur
        _lib.check(self.lib.gpx_set_stream(self.h, C.c_void_p(cur.cuda_stream)), "gpx_set_stream")

    # ---- buffers -------------------------------------------------------------------
    def matrix(self, rows, cols):
        """Uninitialised fp64 rows x cols device matrix; leading dimension padded to 32 doubles."""
        ld = max(32, _round_up(cols, 32))
        buf = self.torch.empty((max(rows, 1), ld), dtype=self.torch.float64, device=self.dev)
        return buf

    def vector(self, n, zero=False):
        f = self.torch.zeros if zero else self.torch.empty
        return f((max(n, 2),), dtype=self.torch.float64, device=self.dev)

    def upload_matrix(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        buf = self.matrix(a.shape[0], a.shape[1])
        buf[: a.shape[0], : a.shape[1]].copy_(self.torch.from_numpy(a))
        return buf

    def upload_vector(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
        buf = self.vector(v.shape[0])
        buf[: v.shape[0]].copy_(self.torch.from_numpy(v))
        return buf

    def download(self, t):
        return t.detach().cpu().numpy()

    def dinv(self, n):
        return self.torch.empty((max(int(self.lib.gpx_dinv_elems(n)), 2),), dtype=self.torch.float64,
                                device=self.dev)


def runtime(index=None):
    import torch
    if index is None:
        index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    with _lock:
        rt = _runtimes.get(index)
        if rt is None:
            rt = Runtime(index)
            _runtimes[index] = rt
    return rt


def on_device(fn):
    """Decorator of the estimator methods that talk to libgpx: one such operation at a time per GPU
    (threads, e.g. joblib's threading backend, are serialised; the lock is re-entrant for nested calls),
    with the handle bound to the caller's current torch stream."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        rt = runtime()
        with rt.lock:
            rt.bind_stream()
            return fn(*args, **kwargs)
    return wrapper


def ptr(t):
    return C.c_void_p(t.data_ptr())
