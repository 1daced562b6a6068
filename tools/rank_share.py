"""One rank's share of a P x Q grid, timed on ONE GPU (development tool).

Runs rank (p, q)'s exact sequence of kernels of the distributed Cholesky (gaussian_process_amd.distributed.
DistributedGP.potrf: the arithmetic of regression.py:36,49 on the 2-D block-cyclic layout) with every collective
replaced by a LOCAL device copy of the byte count that rank would receive: nothing is exchanged, so the values
are garbage after the first foreign tile (the factor is NOT checked here -- the gloo tests do that); what is
measured is what the rank itself has to do:

  gpu_ms        GPU time of the rank's potrf (HIP events around it, queue drained at the end)
  host_ms       host time spent enqueueing it (the Python orchestration: ~30 ctypes / torch calls per panel),
                total and per panel; if host_ms >= gpu_ms the orchestration, not the GPU, bounds the rank
  recv_GB       bytes the rank receives over the factorisation (broadcast pieces it does not own, the other
                members' parts of the transposed all-gather)
  chain_ms      GPU time of the diagonal tiles' potrf, their inverses and the panel solves this rank runs (the
                factorisation's serial chain)

    python tools/rank_share.py [n] [tile] [P] [Q]        (default 65536 512 2 4)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as tdist

from gaussian_process_amd.datasets import synthetic_ard
from gaussian_process_amd.device import runtime
from gaussian_process_amd.distributed import DistributedGP, GpuOps


class _Work:
    def wait(self):
        return True


class SoloDist:
    """torch.distributed's surface as DistributedGP uses it, for ONE rank of a larger world: receives become local
    copies of the same size from a scratch buffer, sends cost nothing."""
    ReduceOp = tdist.ReduceOp

    def __init__(self, rank, device):
        self.rank, self.recv_bytes, self.calls = rank, 0, 0
        self.scratch = torch.full((1 << 26,), 1e-3, dtype=torch.float64, device=device)     # 512 MiB of small numbers

    def get_backend(self, group=None):
        return "nccl"

    def new_group(self, ranks):
        return list(ranks)

    def _fill(self, t):
        flat = t.view(-1)
        for o in range(0, flat.numel(), self.scratch.numel()):
            c = min(self.scratch.numel(), flat.numel() - o)
            flat[o:o + c].copy_(self.scratch[:c])
        self.recv_bytes += 8 * flat.numel()

    def broadcast(self, t, src, group=None, async_op=False):
        self.calls += 1
        if src != self.rank and group is not None and len(group) > 1:
            self._fill(t)
        return _Work()

    def reduce(self, t, dst, op=None, group=None):
        self.calls += 1
        if dst == self.rank and group is not None:
            self.recv_bytes += 8 * t.numel() * (len(group) - 1)

    def all_reduce(self, t, op=None, group=None):
        self.calls += 1
        return _Work()

    def all_gather_into_tensor(self, out, mine, group=None, async_op=False):
        self.calls += 1
        members = len(group)
        parts = out.view(members, -1)
        me = group.index(self.rank)
        for i in range(members):
            if i == me:
                parts[i].copy_(mine)
            else:
                self._fill(parts[i])
        return _Work()

    def barrier(self):
        pass


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    Q = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    world = P * Q
    torch.cuda.set_device(0)
    rt = runtime(0)
    rt.bind_stream()
    X, y, _, beta = synthetic_ard(n, m=4, d=8, seed=0, noise=0.1)
    ranks = [int(r) for r in os.environ.get("RANKS", ",".join(str(r) for r in range(world))).split(",")]
    print("one rank's share of the %d x %d grid: n = %d, tile %d (%d panels); collectives -> local copies" % (P, Q, n, nb, (n + nb - 1) // nb), flush=True)
    worst = None
    for rank in ranks:
        sd = SoloDist(rank, rt.dev)
        gp = DistributedGP(GpuOps(rt), sd, rank, world, block=nb, grid=(P, Q))
        gp.allocate(X, y, "ard", 1.0, beta, 0.1)
        host = {"post": 0.0, "file": 0.0, "bulk": 0.0}
        chain_events = []

        def timed(name, fn):
            def w(*a, **k):
                t0 = time.perf_counter()
                r = fn(*a, **k)
                host[name] += time.perf_counter() - t0
                return r
            return w

        gp._post_panel = timed("post", gp._post_panel)
        gp._file_panel = timed("file", gp._file_panel)
        gp._apply_many = timed("bulk", gp._apply_many)
        ops = gp.ops
        for name in ("potrf_async", "trtri", "trmm_right", "trsm_right"):   # the chain pieces this rank runs, bracketed by events
            fn = getattr(ops, name)

            def w(*a, _fn=fn, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _fn(*a, **k)
                e1.record()
                chain_events.append((e0, e1))
                return r
            setattr(ops, name, w)
        res = {}
        for rep in ("warm", "timed"):
            gp.build_gram()
            torch.cuda.synchronize()
            sd.recv_bytes, sd.calls = 0, 0
            for k_ in host:
                host[k_] = 0.0
            chain_events.clear()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
            gp.potrf()
            t_host = time.perf_counter() - t0      # includes the final info fetch, i.e. the drain of the queue
            e1.record()
            torch.cuda.synchronize()
            res = dict(gpu_ms=e0.elapsed_time(e1), wall_ms=1e3 * t_host, host=dict(host), recv=sd.recv_bytes, calls=sd.calls,
                       chain_ms=sum(a.elapsed_time(b) for a, b in chain_events), grouped=gp.grouped_launches)
        p, q = divmod(rank, Q)
        enq = 1e3 * sum(res["host"].values())
        T = gp.T
        print("rank %d (p=%d, q=%d): gpu %.1f ms | host enqueue %.1f ms = %.3f ms per panel (post %.1f, file %.1f, bulk updates %.1f) | "
              "received %.2f GB in %d collective calls | chain (diagonal potrf + tile inverse + panel solves) %.1f ms | grouped update launches %d | "
              "resident matrix %.2f GB, panels %.2f GB"
              % (rank, p, q, res["gpu_ms"], enq, enq / T, 1e3 * res["host"]["post"], 1e3 * res["host"]["file"],
                 1e3 * res["host"]["bulk"], res["recv"] / 1e9, res["calls"], res["chain_ms"], res["grouped"],
                 gp.resident_bytes["matrix"] / 1e9, gp.resident_bytes["panels"] / 1e9), flush=True)
        if worst is None or res["gpu_ms"] > worst[1]["gpu_ms"]:
            worst = (rank, res, enq)
        del gp, sd
        torch.cuda.empty_cache()
    rank, res, enq = worst
    flops = n ** 3 / 3
    print("slowest rank %d: %.1f ms of GPU work -> if nothing else bound the step, %d ranks would factor at %.1f TFLOP/s "
          "(%.1f per GPU, %.0f %% of the fp64 MFMA peak); its host enqueue time is %.0f %% of its GPU time"
          % (rank, res["gpu_ms"], world, flops / res["gpu_ms"] / 1e9, flops / res["gpu_ms"] / 1e9 / world,
             100 * flops / res["gpu_ms"] / 1e9 / world / 78.6, 100 * enq / res["gpu_ms"]), flush=True)


if __name__ == "__main__":
    main()
