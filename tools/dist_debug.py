"""Single-process (world 1, gloo) trace of the distributed driver: every primitive is followed by
a device sync and a flushed log line, to pin down which call misbehaves for a given shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
from gaussian_process_amd.device import runtime
from gaussian_process_amd.distributed import DistributedGP, GpuOps
from oracle import gp_oracle as orc

n, m, d, block, kind = [int(x) for x in sys.argv[1:5]] + [sys.argv[5]]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
rt = runtime(0); rt.bind_stream()

class Traced:
    def __init__(self, ops):
        self._o = ops
    def __getattr__(self, name):
        f = getattr(self._o, name)
        if not callable(f) or name in ("empty", "zeros"):
            return f
        def g(*a, **k):
            desc = " ".join("%s%s" % (tuple(x.shape), tuple(x.stride())) if hasattr(x, "stride") else repr(x)[:20] for x in a)
            print("  ->", name, desc, flush=True)
            r = f(*a, **k)
            torch.cuda.synchronize()
            print("  <-", name, flush=True)
            return r
        return g

X, y, Xs, _ = orc.synthetic_ard(n, m, d, seed=3)
beta = 0.5 * np.logspace(-1, 0, d) if kind == "ard" else np.array([0.7])
gp = DistributedGP(Traced(GpuOps(rt)), dist, rank, world, block=block)
_b = gp._bcast
def traced_bcast(t, src, async_op=False):
    print("  bcast post", tuple(t.shape), "src", src, "async", async_op, flush=True)
    w = _b(t, src, async_op)
    if w is None:
        print("  bcast done", flush=True)
        return w
    class W:
        def wait(self_):
            print("  bcast wait...", flush=True); w.wait(); torch.cuda.synchronize(); print("  bcast waited", flush=True)
    return W()
gp._bcast = traced_bcast
print("fit", flush=True)
lml = gp.fit(X, y, kind, 1.3, beta, 0.05)
print("lml", lml, flush=True)
mean, cov = gp.predict(Xs)
print("predict done", flush=True)
print("res", gp.residual(), flush=True)
lml_o, _, Lo = orc.gpr_lml(kind, X, y, 1.3, beta, 0.05)
print("lml rel", abs(lml - lml_o) / abs(lml_o))
