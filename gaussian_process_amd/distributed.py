"""Multi-GPU exact GP regression: 2-D block-cyclic, SHARDED Cholesky and triangular solves over
``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

One process per GPU.  The reference has no distributed code at all; what is distributed here is
the arithmetic of regression.py:31-76 (Gram, ``cholesky``, the triangular solves, LML, its gradient,
predictive mean / covariance), with the same kernels as the single-GPU path (libgpx) doing all
local work.

Layout (index maps in grid2d.py): a P x Q process grid, rank = p * Q + q; the
matrix is cut into nb x nb tiles and tile (I, J), I >= J, lives on process (I mod P, J mod Q).  A
process keeps one contiguous slab per tile column it owns (its tiles of that column from the
diagonal down): the resident matrix is the lower triangle / (P Q), nothing is replicated.

Factorisation, step k (right-looking, look-ahead 1, delayed trailing updates):
  1. the diagonal owner factors tile (k, k) (``gpx_potrf_async``) and broadcasts it with its
     128 x 128 diagonal-block inverses down its process COLUMN (sub-communicator of P ranks);
  2. the P processes of that column solve their rows of the panel (``gpx_trsm_right``);
  3. each broadcasts its rows along its process ROW (sub-communicator of Q ranks): every process
     then holds L[I, k] for its tile rows I -- the left operand of its trailing update;
  4. transposed exchange: inside every process column the ranks all-gather the tiles L[J, k] with
     J mod Q == q: the right operand (J runs over the process's tile COLUMNS);
  5. local update A[I, J] -= L[I, k] L[J, k]^T.  The last G panels stay in two ring buffers (by local
     row / local column), so a tile column absorbs several panels in ONE GEMM with a long K.
Per step a process receives (1/P + 1/Q) of the panel, not all of it.

After the factorisation L never moves.  ``L^-1 B`` runs right-looking with fan-in: the ranks of
process column (k mod Q) subtract L[I, k] X[k] from the partial sums of their rows, the partial sums of
row k are reduced along the process row to the diagonal owner, which solves X[k] and broadcasts it
down its column.  ``L^-T B`` is the mirror image (reduce down the column, broadcast along the row).
Only right-hand-side tiles travel: (n/P + n/Q) m doubles per solve and rank.  alpha / LML use m = 1,
prediction m test points, the LML gradient slabs of identity columns (Ky^-1 by tile-row slabs, never
whole) contracted on the fly by ``gpx_grad_trace_slab``; scalars are all-reduced.
"""
import ctypes as C
import os
from contextlib import contextmanager

import numpy as np

from . import _lib
from .grid2d import Grid2D, default_grid


def default_block(world):
    """Tile size nb per world size: wide tiles keep the local GEMMs efficient, narrow ones shorten the
    factorisation chain (its length is n / nb steps whatever the number of ranks)."""
    return {1: 2048, 2: 1024, 4: 1024, 8: 512}.get(world, 1024)


class _Done:
    def wait(self):
        return True


class GpuOps:
    """Local kernels through the C ABI on torch device tensors (views carry pointer + stride)."""

    def __init__(self, rt):
        self.rt = rt
        self.lib = rt.lib
        self.h = rt.h
        self.torch = rt.torch
        self.device = rt.dev

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def empty(self, *shape, dtype=None):
        return self.torch.empty(shape, dtype=dtype or self.torch.float64, device=self.device)

    def zeros(self, *shape, dtype=None):
        return self.torch.zeros(shape, dtype=dtype or self.torch.float64, device=self.device)

    def gram(self, kind, sigma, beta, Xr, Xc, out, diag_add):
        barr, bp = _lib.host_doubles(beta)
        _lib.check(self.lib.gpx_gram(self.h, kind, self._p(Xr), Xr.shape[0], Xr.stride(0), self._p(Xc),
                                     Xc.shape[0], Xc.stride(0), Xr.shape[1], float(sigma), bp, float(diag_add),
                                     self._p(out), out.stride(0), _lib.GPX_FULL), "gpx_gram")

    def potrf_async(self, Akk, dinv, info, info_offset):
        _lib.check(self.lib.gpx_potrf_async(self.h, Akk.shape[0], self._p(Akk), Akk.stride(0), self._p(dinv),
                                            self._p(info), int(info_offset)), "gpx_potrf_async")

    def trsm_right(self, Xp, Lkk, dinv):
        _lib.check(self.lib.gpx_trsm_right(self.h, Xp.shape[0], Xp.shape[1], self._p(Lkk), Lkk.stride(0),
                                           self._p(dinv), self._p(Xp), Xp.stride(0)), "gpx_trsm_right")

    def gemm(self, ta, tb, M, N, K, alpha, A, B, beta, Cm):
        _lib.check(self.lib.gpx_gemm(self.h, ta, tb, M, N, K, float(alpha), self._p(A), A.stride(0), self._p(B),
                                     B.stride(0), float(beta), self._p(Cm), Cm.stride(0)), "gpx_gemm")

    def gemm_colblocks(self, M, N, K, alpha, A, B, beta, store, nb, row_base, blk_off, blk_row0):
        """store[slab of column block c] = alpha * A B_c^T + beta * (same), every nb-wide column block c of the N
        columns in ONE launch (gpx_gemm_colblocks; the tables are device tensors, see allocate())."""
        _lib.check(self.lib.gpx_gemm_colblocks(self.h, M, N, K, float(alpha), self._p(A), A.stride(0), self._p(B),
                                               B.stride(0), float(beta), self._p(store), nb, int(row_base),
                                               self._p(blk_off), self._p(blk_row0)), "gpx_gemm_colblocks")

    def trsm_left(self, trans, L, dinv, B):
        _lib.check(self.lib.gpx_trsm(self.h, trans, L.shape[0], B.shape[1], self._p(L), L.stride(0),
                                     self._p(dinv), self._p(B), B.stride(0)), "gpx_trsm")

    def trtri(self, W, dinv):
        """W <- W^-1 in place (lower; W's blocks above the diagonal blocks are scratch)."""
        _lib.check(self.lib.gpx_trtri(self.h, W.shape[0], self._p(W), W.stride(0), self._p(dinv)), "gpx_trtri")

    def trmm(self, trans, W, B, Out):
        """Out = W B (trans 0) or W^T B (trans 1), W lower triangular."""
        _lib.check(self.lib.gpx_trmm(self.h, trans, W.shape[0], B.shape[1], self._p(W), W.stride(0), self._p(B),
                                     B.stride(0), self._p(Out), Out.stride(0)), "gpx_trmm")

    def trmm_right(self, X, W, Out):
        """Out = X W^T, W lower triangular (the panel solve with a diagonal tile's explicit inverse)."""
        _lib.check(self.lib.gpx_trmm_right(self.h, X.shape[0], W.shape[0], self._p(W), W.stride(0), self._p(X), X.stride(0),
                                           self._p(Out), Out.stride(0)), "gpx_trmm_right")

    def logdet(self, L):
        out = C.c_double()
        _lib.check(self.lib.gpx_logdet(self.h, L.shape[0], self._p(L), L.stride(0), C.byref(out)), "gpx_logdet")
        return out.value

    def dot(self, n, x, y):
        out = C.c_double()
        _lib.check(self.lib.gpx_dot(self.h, n, self._p(x), self._p(y), C.byref(out)), "gpx_dot")
        return out.value

    def gemv_t(self, A, x, y):
        _lib.check(self.lib.gpx_gemv(self.h, 1, A.shape[0], A.shape[1], self._p(A), A.stride(0), self._p(x),
                                     self._p(y)), "gpx_gemv")

    def gemv_n(self, A, x, y):
        _lib.check(self.lib.gpx_gemv(self.h, 0, A.shape[0], A.shape[1], self._p(A), A.stride(0), self._p(x),
                                     self._p(y)), "gpx_gemv")

    def gemv_acc(self, trans, alpha, A, x, beta, y):
        """y = alpha * op(A) x + beta * y for contiguous vectors x, y."""
        _lib.check(self.lib.gpx_gemv_acc(self.h, trans, A.shape[0], A.shape[1], float(alpha), self._p(A), A.stride(0),
                                         self._p(x), float(beta), self._p(y)), "gpx_gemv_acc")

    def colsumsq(self, V, base, var):
        """var[j] = base - sum_i V[i][j]^2."""
        _lib.check(self.lib.gpx_colsumsq_sub(self.h, V.shape[0], V.shape[1], self._p(V), V.stride(0), float(base),
                                             self._p(var)), "gpx_colsumsq_sub")

    def grad_trace_slab(self, kind, sigma, beta, X, d, alpha, slab, row_off, col_off):
        barr, bp = _lib.host_doubles(beta)
        out = (C.c_double * (2 + len(barr)))()
        _lib.check(self.lib.gpx_grad_trace_slab(self.h, kind, self._p(X), X.shape[0], X.stride(0), d, float(sigma),
                                                bp, self._p(alpha), self._p(slab), slab.stride(0), int(row_off),
                                                int(col_off), slab.shape[0], slab.shape[1],
                                                C.cast(out, C.POINTER(C.c_double))), "gpx_grad_trace_slab")
        return np.array(list(out))

    # ---- a second stream for work that should overlap the main stream (the diagonal-tile potrf of
    # the look-ahead column is a chain of one-workgroup kernels; the column update fills the other CUs)
    def side_begin(self):
        t = self.torch
        if not hasattr(self, "_side"):
            self._side = t.cuda.Stream(device=self.device)
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self.device))
        self._side.wait_event(ev)
        _lib.check(self.lib.gpx_set_stream(self.h, C.c_void_p(self._side.cuda_stream)), "gpx_set_stream")

    def side_end(self):
        """Back to the main stream; returns a token for side_join()."""
        t = self.torch
        ev = t.cuda.Event()
        ev.record(self._side)
        _lib.check(self.lib.gpx_set_stream(self.h, C.c_void_p(t.cuda.current_stream(self.device).cuda_stream)),
                   "gpx_set_stream")
        return ev

    def side_join(self, ev):
        self.torch.cuda.current_stream(self.device).wait_event(ev)

    @contextmanager
    def forward_stream(self):
        """Stream context for the copies and collectives that only FORWARD a received panel (the
        transposed exchange): ordered after what the compute stream has enqueued so far, but not
        behind what it enqueues next (the bulk trailing updates)."""
        t = self.torch
        if not hasattr(self, "_fwd"):
            self._fwd = t.cuda.Stream(device=self.device)
        ev = t.cuda.Event()
        ev.record(t.cuda.current_stream(self.device))
        self._fwd.wait_event(ev)
        with t.cuda.stream(self._fwd):
            yield

    def sync(self):
        self.torch.cuda.synchronize(self.device)


class DistributedGP:
    """Exact GP regression over the ranks of a torch.distributed process group, 2-D block-cyclic."""

    def __init__(self, ops, dist, rank, world, block=2048, grid=None, update_k=None, grad_block=None, group=None,
                 force_collectives=False, many_rhs=None):
        assert block % _lib.GPX_NB == 0, "block must be a multiple of 128"
        self.ops, self.dist, self.rank, self.world, self.block, self.group = ops, dist, rank, world, block, group
        self.torch = ops.torch
        self.P, self.Q = grid if grid is not None else default_grid(world)
        assert self.P * self.Q == world, "grid %dx%d does not match world size %d" % (self.P, self.Q, world)
        self.p, self.q = divmod(rank, self.Q)
        # K the delayed trailing updates aim for: a tile column absorbs update_k / block panels in one
        # GEMM (K = block alone runs 15-20 % slower on the MFMA kernel and rewrites the column every step)
        # (K = 4096 and 8192 run alike on one rank -- potrf 1474 against 1472 ms at n = 65536 -- and the ring of
        # pending panels is half as large)
        # Two ranks (2 x 1): tile 1024 and K = 2048 keep the rings and the staging below half of the rank's share of
        # the matrix (the column ring holds ALL tile columns on a P x 1 grid).
        self.update_k = update_k or {1: 4096}.get(world, 2048)
        # columns of Ky^-1 per gradient slab (each slab is one forward + one backward solve)
        self.grad_block = grad_block or 4096
        if many_rhs:
            self.MANY_RHS = many_rhs
        # Panel solve with the diagonal tile's explicit inverse (one triangular-aware MFMA product, written straight
        # into the ring of pending panels) instead of the recursion over the tile's 128-blocks; development switch
        self.panel_inverse = os.environ.get("GPX_DIST_PANEL_INV", "1") != "0" and hasattr(ops, "trmm_right")
        # development switch: one launch per tile column (round 2) instead of grouped trailing updates, for A/B runs
        self.group_updates = os.environ.get("GPX_DIST_GROUP", "1") != "0"
        # development switch: the round-2 placement of the diagonal-tile potrf (side stream) for A/B runs
        self.side_potrf = os.environ.get("GPX_DIST_SIDE_POTRF", "0") == "1" and hasattr(ops, "side_begin")
        # which communicators carry traffic: a dimension of extent 1 needs none.  force_collectives
        # issues them anyway (groups of one rank): the world-size-1 rehearsal of the RCCL branch.
        self.rc = self.Q > 1 or force_collectives          # row sub-communicator (Q ranks)
        self.cc = self.P > 1 or force_collectives          # column sub-communicator (P ranks)
        self.wc = world > 1 or force_collectives           # world group
        self._gloo = self.wc and dist.get_backend(group) == "gloo"
        self.row_group = self.col_group = None
        # every rank creates every sub-communicator, in the same order
        for p in range(self.P):
            g = dist.new_group([p * self.Q + q for q in range(self.Q)]) if self.rc else None
            if p == self.p:
                self.row_group = g
        for q in range(self.Q):
            g = dist.new_group([p * self.Q + q for p in range(self.P)]) if self.cc else None
            if q == self.q:
                self.col_group = g
        # Touch every communicator once, rows first, then columns, the same order on every rank: RCCL builds a
        # communicator at its first collective (a host-side rendezvous of its members); doing that here keeps
        # it out of the factorisation, where the first uses are asynchronous and interleaved with kernels.
        if self.wc:
            one = self.torch.ones(1, dtype=self.torch.float64, device=getattr(ops, "device", "cpu"))
            for grp in (self.row_group if self.rc else None, self.col_group if self.cc else None):
                if grp is not None:
                    self._all_reduce(one, group=grp)
            self._all_reduce(one)
        self.n = 0
        self._ws = {}
        self._winv = {}

    # ------------------------------------------------------------------ collectives
    def _staged(self, t):
        return t.is_cuda and self._gloo

    def _bcast(self, t, src, group, async_op=False):
        """Broadcast a contiguous buffer inside `group` (src = global rank).  RCCL: asynchronous on
        RCCL's stream, ordered after the kernels already enqueued on the current stream; wait() makes
        the current stream wait, not the host.  gloo with device tensors (rehearsals with several ranks
        on one GPU): staged through the host and completed before returning."""
        assert t.is_contiguous()       # RCCL needs it; checked on every backend so the gloo tests cover it
        if self._staged(t):
            host = t.cpu()
            self.dist.broadcast(host, src=src, group=group)
            if self.rank != src:
                t.copy_(host)
            return _Done()
        w = self.dist.broadcast(t, src=src, group=group, async_op=async_op)
        return w if async_op else _Done()

    def _reduce(self, t, dst, group):
        """Sum `t` over `group` into rank dst's `t` (the other ranks' buffers are scratch afterwards)."""
        assert t.is_contiguous()
        if self._staged(t):
            host = t.cpu()
            self.dist.reduce(host, dst=dst, op=self.dist.ReduceOp.SUM, group=group)
            if self.rank == dst:
                t.copy_(host)
            return
        self.dist.reduce(t, dst=dst, op=self.dist.ReduceOp.SUM, group=group)

    def _all_reduce(self, t, op=None, group=None):
        op = op or self.dist.ReduceOp.SUM
        if not self.wc:
            return t
        if self._staged(t):
            host = t.cpu()
            self.dist.all_reduce(host, op=op, group=group or self.group)
            t.copy_(host)
            return t
        self.dist.all_reduce(t, op=op, group=group or self.group)
        return t

    def _all_gather(self, out, mine, group, async_op=False):
        """out (members x len(mine)) <- every member's `mine`, in group-rank order."""
        if self._staged(mine):
            assert out.is_contiguous() and mine.is_contiguous()
            host = mine.cpu()
            parts = [self.torch.empty_like(host) for _ in range(out.shape[0])]
            self.dist.all_gather(parts, host, group=group)
            out.copy_(self.torch.stack(parts))
            return _Done()
        if self._gloo or not mine.is_cuda:
            assert out.is_contiguous() and mine.is_contiguous()
            self.dist.all_gather([out[i] for i in range(out.shape[0])], mine, group=group)
            return _Done()
        assert out.is_contiguous() and mine.is_contiguous()
        w = self.dist.all_gather_into_tensor(out.view(-1), mine, group=group, async_op=async_op)
        return w if async_op else _Done()

    @contextmanager
    def _forwarding(self):
        if self._gloo or not self.cc or not hasattr(self.ops, "forward_stream"):
            yield
        else:
            with self.ops.forward_stream():
                yield

    def _grank(self, p, q):
        return p * self.Q + q

    # ------------------------------------------------------------------ fit
    def allocate(self, X, y, kind, sigma, beta, noise):
        """Upload the (replicated, 8 n d bytes) inputs and allocate everything fit/predict keep
        resident: this process's slabs of the matrix and the panel workspaces."""
        ops, t = self.ops, self.torch
        n, d = X.shape
        self.n, self.d = n, d
        self.set_theta(kind, sigma, beta, noise)
        nb = min(self.block, ((n + 127) // 128) * 128)
        g = self.grid = Grid2D(n, nb, self.P, self.Q)
        self.nb, self.T = nb, g.T
        p, q = self.p, self.q
        X = np.ascontiguousarray(X, dtype=np.float64)
        ldx = max(32, ((d + 31) // 32) * 32)
        self.X = ops.zeros(n, ldx)
        self.X[:, :d].copy_(t.from_numpy(X))
        # the rows of X of this process row's tile rows, in local order (left operand of the Gram slabs)
        nlr, nlc = g.nlr(p), g.nlc(q)
        Xr = np.zeros((max(nlr, 1) * nb, d))
        for I in g.local_rows(p):
            Xr[g.lrow(I): g.lrow(I) + g.tile_h(I)] = X[I * nb: I * nb + g.tile_h(I)]
        self.Xrow = ops.zeros(max(nlr, 1) * nb, ldx)
        self.Xrow[:, :d].copy_(t.from_numpy(Xr))
        self.y = ops.empty(max(n, 2))
        self.y[:n].copy_(t.from_numpy(np.ascontiguousarray(y, dtype=np.float64)))
        # ---- the resident matrix: one slab (whole tiles high, nb wide) per owned tile column
        self.slab_rows = {J: g.slab_rows(p, J) for J in g.local_cols(q)}
        self.slab_tiles = {J: len(g.local_rows(p, J)) for J in g.local_cols(q)}
        total = sum(self.slab_tiles[J] * nb * nb for J in self.slab_tiles)
        self.store = ops.empty(max(total, 2))
        self.slabs, off = {}, 0
        for J in g.local_cols(q):
            cnt = self.slab_tiles[J] * nb * nb
            self.slabs[J] = self.store[off: off + cnt].view(self.slab_tiles[J] * nb, nb)
            off += cnt
        # column-block tables of the grouped trailing update (gpx_gemm_colblocks), by LOCAL tile column: element
        # (local row r, column j) of tile column J lives at store[cb_off + r * nb + j] for r >= cb_row0
        offs, row0s, off = [], [], 0
        for J in g.local_cols(q):
            base = g.lrow(g.first_at_or_after(J, p, self.P))
            live = self.slab_rows[J] > 0
            offs.append(off - base * nb if live else 0)
            # bit 0: the slab's first tile is the diagonal tile (this process row owns it) -- the launch skips the
            # 128-tiles strictly above its diagonal.  No rows: every tile of the block is skipped.
            row0s.append((base | (1 if J % self.P == p else 0)) if live else (max(nlr, 1) + 1) * nb)
            off += self.slab_tiles[J] * nb * nb
        self.cb_off = t.tensor(offs or [0], dtype=t.int64).to(self.store.device)
        self.cb_row0 = t.tensor(row0s or [0], dtype=t.int32).to(self.store.device)
        dper = (nb // 128) * 128 * 128
        diag = g.diag_tiles(p, q)
        self.dinv_store = ops.empty(max(len(diag) * dper, 2))
        self.dinv = {k: self.dinv_store[i * dper: i * dper + ((g.tile_h(k) + 127) // 128) * 128 * 128]
                     for i, k in enumerate(diag)}
        self.info = ops.zeros(64, dtype=t.int32)
        # ---- panel workspaces: ring buffers of the last G panels, by local row and by local column
        self.g_upd = max(1, self.update_k // nb)
        self.G = self.g_upd + 2
        self.RB = ops.empty(max(nlr, 1) * nb, self.G * nb)
        # On a 1 x 1 grid the panel by local row and the panel by local column are the same tiles in the same
        # places: one ring serves as both operands of the trailing update.
        self.cb_is_rb = self.P == 1 and self.Q == 1
        self.CB = self.RB if self.cb_is_rb else ops.empty(max(nlc, 1) * nb, self.G * nb)
        # Staging of the collectives, ONE buffer each (round 2 kept two, by panel parity).  Panel k + 1's collectives
        # are issued after _file_panel(k) has enqueued its copies out of the staging buffers, and every collective
        # (and the forwarding stream) starts by waiting for what the compute stream had enqueued at that moment, so a
        # buffer is never rewritten before its last reader -- on the sender as on the receivers, each of which
        # orders its own side the same way.  Allocated only for the communicators that carry traffic.
        self.rowstg = ops.empty(max(nlr, 1) * nb * nb) if self.rc else None
        cmax = max([g.transposed_max(q, k) for k in range(g.T)] + [1])
        # flat, so that the (P x count) view of any panel is contiguous
        self.tsend = ops.empty(cmax * nb * nb) if self.cc else None
        self.trecv = ops.empty(self.P * cmax * nb * nb) if self.cc else None
        self.dstg = ops.empty(nb * nb + dper) if self.cc else None
        self._ws = {}
        numel = lambda b: 0 if b is None else b.numel()
        self.resident_bytes = dict(matrix=8 * total,
                                   panels=8 * (self.RB.numel() + (0 if self.cb_is_rb else self.CB.numel()) + numel(self.rowstg)
                                               + numel(self.tsend) + numel(self.trecv) + numel(self.dstg)),
                                   # explicit inverses of the owned diagonal tiles (panel solves, many-right-hand-side
                                   # tile solves): 1 / T of the matrix, allocated as the factorisation reaches them
                                   tile_inverses=8 * sum(g.tile_h(k) * max(32, ((g.tile_h(k) + 31) // 32) * 32) for k in diag),
                                   lower_over_world=8 * g.lower_elems() // self.world)

    def set_theta(self, kind, sigma, beta, noise):
        """Hyper-parameters of the next build_gram(): kernel kind ("rbf" / "ard" or the C ABI's id),
        kernel amplitude, beta (1 value for RBF, d for ARD), noise variance."""
        self.kind = _lib.GPX_KIND[kind] if isinstance(kind, str) else int(kind)
        self.sigma, self.noise = float(sigma), float(noise)
        self.beta = np.asarray(beta, dtype=np.float64).reshape(-1)
        if self.kind == _lib.GPX_KIND["rbf"]:
            self.beta = self.beta[:1]

    def upload_test(self, Xs):
        t = self.torch
        Xs_d = self.ops.zeros(max(Xs.shape[0], 1), self.X.shape[1])
        Xs_d[: Xs.shape[0], : self.d].copy_(t.from_numpy(np.ascontiguousarray(Xs, dtype=np.float64)))
        return Xs_d[: Xs.shape[0]]

    def fit(self, X, y, kind, sigma, beta, noise):
        """Gram + distributed Cholesky + alpha + LML.  X (n, d), y (n,) host arrays, identical on
        every rank.  Returns the LML; the sharded factor stays resident for predict() / grad()."""
        self.allocate(X, y, kind, sigma, beta, noise)
        self.build_gram()
        self.potrf()
        return self.solve_lml()

    def build_gram(self):
        """Each process fills its own slabs from the replicated X (kernels.py:77-79 + the noise of
        regression.py:48 on the diagonal tiles): no communication."""
        g, d, nb = self.grid, self.d, self.nb
        for J, slab in self.slabs.items():
            rows = self.slab_rows[J]
            if rows == 0:
                continue
            I0 = g.first_at_or_after(J, self.p, self.P)
            wJ = g.tile_h(J)
            self.ops.gram(self.kind, self.sigma, self.beta, self.Xrow[g.lrow(I0): g.lrow(I0) + rows, :d],
                          self.X[J * nb: J * nb + wJ, :d], slab[:rows, :wJ], self.noise if I0 == J else 0.0)

    # ------------------------------------------------------------------ factorisation
    def _update(self, J, s0, cnt, r0, r1):
        """slab_J[r0:r1] -= RB[rows, slots] CB[tile column J, slots]^T: `cnt` consecutive ring slots
        starting at s0 applied to rows [r0, r1) of tile column J in ONE GEMM with K = cnt * nb."""
        if r1 <= r0 or cnt <= 0:
            return
        g, nb = self.grid, self.nb
        base = g.lrow(g.first_at_or_after(J, self.p, self.P))
        wJ = g.tile_h(J)
        self.ops.gemm(0, 1, r1 - r0, wJ, cnt * nb, -1.0, self.RB[base + r0: base + r1, s0 * nb:(s0 + cnt) * nb],
                      self.CB[g.lcol(J): g.lcol(J) + wJ, s0 * nb:(s0 + cnt) * nb], 1.0, self.slabs[J][r0:r1, :wJ])

    def _apply(self, J, upto, r0=0, r1=None, commit=True):
        """Apply the filed panels [applied[J], upto) to rows [r0, r1) of tile column J."""
        a, G = self._applied[J], self.G
        r1 = self.slab_rows[J] if r1 is None else r1
        while a < upto:
            s0 = a % G
            cnt = min(upto - a, G - s0)
            self._update(J, s0, cnt, r0, r1)
            a += cnt
        if commit:
            self._applied[J] = upto

    def _apply_many(self, cols, upto):
        """Bring the tile columns `cols` (ascending) up to panel `upto`.  Runs of consecutive local columns with
        the same pending range share their left operand (the rows of RB below the first column's diagonal tile)
        and go out as ONE launch with N = (columns) * nb (gpx_gemm_colblocks: every column block lands in its
        own slab, the tiles above a slab's diagonal tile are skipped) instead of one skinny N = nb product per
        column -- at nb = 512 those ran at 56-58 TFLOP/s where a wide launch reaches 68-70."""
        g, nb, G = self.grid, self.nb, self.G
        grouped = self.group_updates and hasattr(self.ops, "gemm_colblocks")
        i = 0
        while i < len(cols):
            J0 = cols[i]
            j = i + 1
            while (grouped and j < len(cols) and cols[j] == cols[j - 1] + self.Q
                   and self._applied[cols[j]] == self._applied[J0]):
                j += 1
            members = cols[i:j]
            i = j
            run = [J for J in members if self.slab_rows[J] > 0]
            if len(run) < 2 or self._applied[J0] >= upto:
                for J in members:
                    self._apply(J, upto)
                continue
            Ja = run[0]
            base = g.lrow(g.first_at_or_after(Ja, self.p, self.P))
            M = self.slab_rows[Ja]
            N = sum(g.tile_h(J) for J in run)
            lc0 = Ja // self.Q
            a = self._applied[Ja]
            while a < upto:
                s0 = a % G
                cnt = min(upto - a, G - s0)
                self.ops.gemm_colblocks(M, N, cnt * nb, -1.0, self.RB[base: base + M, s0 * nb:(s0 + cnt) * nb],
                                        self.CB[g.lcol(Ja): g.lcol(Ja) + N, s0 * nb:(s0 + cnt) * nb], 1.0,
                                        self.store, nb, base, self.cb_off[lc0:], self.cb_row0[lc0:])
                self.grouped_launches += 1
                a += cnt
            for J in members:
                self._applied[J] = upto

    def _post_panel(self, k):
        """Everything that produces panel k and starts it on its way: (column owners) bring tile
        column k up to date, (diagonal owner) factor the diagonal tile and broadcast it down the
        column, (column owners) solve their rows, then -- every process -- the row broadcast and the
        transposed exchange.  Returns the handles _file_panel(k) waits for."""
        g, ops, nb, P, Q, p, q = self.grid, self.ops, self.nb, self.P, self.Q, self.p, self.q
        pk, qk = k % P, k % Q
        has_panel = k < g.T - 1
        wk = g.tile_h(k)
        below = None
        filed_rows = False
        if q == qk:
            slab, rows = self.slabs[k], self.slab_rows[k]
            mine = p == pk
            h_diag, dst = None, None
            if mine:
                # The diagonal tile first, on the compute stream itself, AHEAD of the update of the rows below it:
                # the tile's potrf is the factorisation's critical chain (every rank of the column waits for it),
                # and its 128 x 128 leaves need a whole CU's LDS each.  (Round 2 ran it on a side stream beside
                # the update of the rows below; with every CU holding two 64-KiB GEMM workgroups a leaf waited
                # for a CU to drain -- 239 us per leaf on average against 30 us alone, 9 % of the whole potrf.)
                self._apply(k, k, 0, wk, commit=False)
                if self.side_potrf:
                    ops.side_begin()
                    ops.potrf_async(slab[:wk, :wk], self.dinv[k], self.info, k * nb)
                    token = ops.side_end()
                    self._apply(k, k, wk, rows)
                    ops.side_join(token)
                else:
                    ops.potrf_async(slab[:wk, :wk], self.dinv[k], self.info, k * nb)
            Wkk = None
            if has_panel and self.panel_inverse and mine:
                # W_kk = L_kk^-1 right behind the tile's potrf (level-by-level trtri on a copy, kept for the
                # many-right-hand-side tile solves of predict / grad): the panel solve below is then ONE product
                Wkk = self._tile_inverse(k)
            if has_panel and self.cc:
                # the broadcast down the process column leaves as soon as the tile is factored; the update of
                # the rows below (next) runs under it
                dst = self.dstg
                if mine:
                    if self.panel_inverse:
                        dst[: nb * nb].view(nb, nb).copy_(Wkk[:, :nb])
                    else:
                        dst[: nb * nb].view(nb, nb).copy_(slab[:nb])
                        dst[nb * nb:].copy_(self.dinv[k])
                h_diag = self._bcast(dst, self._grank(pk, qk), self.col_group, async_op=True)
            if not (mine and self.side_potrf):
                self._apply(k, k, wk if mine else 0, rows)
            if has_panel:
                Lkk, dk = (slab[:nb], self.dinv[k]) if mine else (None, None)
                if h_diag is not None:
                    h_diag.wait()
                    if not mine:
                        Lkk, dk = dst[: nb * nb].view(nb, nb), dst[nb * nb:]
                        Wkk = Lkk                     # (panel_inverse: the payload is the inverse)
                below = slab[nb:rows] if mine else slab[:rows]
                if below.shape[0] > 0:
                    if self.panel_inverse:
                        # solved rows straight into ring slot k (free since panel k - G was applied everywhere one
                        # step ago), then back into the slab, which keeps the factor and feeds the row broadcast
                        I1 = g.first_at_or_after(k + 1, p, P)
                        slot = k % self.G
                        ring = self.RB[g.lrow(I1): g.lrow(I1) + below.shape[0], slot * nb:(slot + 1) * nb]
                        ops.trmm_right(below, Wkk[:nb, :nb], ring)
                        below.copy_(ring)
                        filed_rows = True
                    else:
                        ops.trsm_right(below, Lkk, dk)
        if not has_panel:
            return None
        # Row broadcast: skipped by a whole process row that has no tile rows below k.  The transposed
        # exchange is joined by every rank of a process column that needs any tile, also by ranks whose
        # own row piece is empty.
        R = g.rows_from(p, k + 1)
        ntiles = len(g.local_rows(p, k + 1))
        piece, h_row = None, _Done()
        if R > 0:
            if q == qk:
                start = nb if p == pk else 0
                piece = self.slabs[k][start: start + ntiles * nb]          # whole tiles (slabs are padded)
            else:
                piece = self.rowstg[: ntiles * nb * nb].view(ntiles * nb, nb)
            if self.rc:
                h_row = self._bcast(piece.reshape(-1)[: R * nb], self._grank(p, qk), self.row_group, async_op=True)
        out = dict(R=R, piece=piece, h_row=h_row, cnt_max=g.transposed_max(q, k), h_T=_Done(), filed_rows=filed_rows)
        if out["cnt_max"] > 0 and self.cc:
            t0, step, cnt = g.transposed_pick(p, q, k)
            cm = out["cnt_max"]
            with self._forwarding():
                h_row.wait()
                send = self.tsend
                if cnt > 0:
                    send[: cnt * nb * nb].view(cnt, nb, nb).copy_(piece.view(ntiles, nb, nb)[t0::step][:cnt])
                out["h_T"] = self._all_gather(self.trecv[: P * cm * nb * nb].view(P, cm * nb * nb),
                                              send[: cm * nb * nb], self.col_group, async_op=True)
        return out

    def _file_panel(self, k, h):
        """Wait for panel k and file it into ring slot k % G: rows by local row, the transposed tiles by
        local column."""
        g, nb, P, Q, p, q = self.grid, self.nb, self.P, self.Q, self.p, self.q
        if h is None:
            return
        slot = k % self.G
        cols = slice(slot * nb, (slot + 1) * nb)
        R = h["R"]
        if R > 0:
            I1 = g.first_at_or_after(k + 1, p, P)
            h["h_row"].wait()
            if not h["filed_rows"]:          # (the panel solve of the owning process column wrote the ring itself)
                self.RB[g.lrow(I1): g.lrow(I1) + R, cols].copy_(h["piece"][:R])
        if h["cnt_max"] == 0:
            return
        h["h_T"].wait()
        if self.cb_is_rb:
            return                      # 1 x 1 grid: the rows just filed are the column tiles as well
        nlc = self.CB.shape[0] // nb
        CBt = self.CB.view(nlc, nb, self.G * nb)
        stride = g.L // Q
        for src in range(P):
            t0, step, cnt = g.transposed_pick(src, q, k)
            if cnt == 0:
                continue
            first = g.local_rows(src, k + 1)[t0]                       # a tile index I with I mod Q == q
            if self.cc:
                cm = h["cnt_max"]
                tiles = self.trecv[: P * cm * nb * nb].view(P, cm * nb * nb)[src, : cnt * nb * nb].view(cnt, nb, nb)
            else:
                tiles = h["piece"].view(-1, nb, nb)[t0::step][:cnt]
            CBt[first // Q:: stride][:cnt, :, cols].copy_(tiles)

    def potrf(self):
        """Distributed right-looking Cholesky (regression.py:36,49) with look-ahead 1 and delayed
        trailing updates.  Returns LAPACK info (0, or the order of the first leading minor that is not
        positive definite), the same value on every rank."""
        g, t, T, G = self.grid, self.torch, self.T, self.G
        self.info.zero_()
        self._winv = {}                      # tile inverses belong to the previous factor
        self._applied = {J: 0 for J in self.slabs}
        self.grouped_launches = 0            # diagnostics: trailing updates issued as grouped launches
        handles = self._post_panel(0)
        for k in range(T - 1):
            self._file_panel(k, handles)
            # look-ahead: tile column k+1 is brought up to date, factored and sent before the bulk
            handles = self._post_panel(k + 1)
            # The process column that factors panel k+2 next step enqueues only that column's catch-up
            # now: its other updates would sit in the stream ahead of the chain it is about to run, and
            # lose nothing by waiting one step (they come back with a longer K).
            chain_next = k + 2 < T and (k + 2) % self.Q == self.q
            due = []
            for J in g.local_cols(self.q, k + 2):
                pending = k + 1 - self._applied[J]
                if J == k + 2 or pending >= G - 1 or (pending >= self.g_upd and not chain_next):
                    due.append(J)
            self._apply_many(due, k + 1)
        # LAPACK info: smallest failing leading minor over all ranks (0 = success)
        big = 2 ** 30
        loc = int(self.info[0].item())
        val = t.tensor([loc if loc > 0 else big], dtype=t.int64, device=self.store.device)
        val = self._all_reduce(val, self.dist.ReduceOp.MIN)
        self.info_value = 0 if int(val.item()) >= big else int(val.item())
        return self.info_value

    # ------------------------------------------------------------------ triangular solves
    def _buffer(self, name, rows, cols):
        buf = self._ws.get(name)
        if buf is None or buf.shape[0] < rows or buf.shape[1] != cols:
            buf = self._ws[name] = self.ops.empty(max(rows, 1), cols)
        return buf

    def rhs_buffer(self, m, zero=True, vector=False):
        """Right-hand-side / solution container of the solves: (local tile rows * nb) x ld, ld = m
        rounded up to even (vector=True: m = 1 and ld = 1).  Tile row k of B lives in rows [lrow(k), lrow(k) + h_k) of the DIAGONAL
        OWNER of k; the other rows are the fan-in workspace (partial sums / received solution tiles)."""
        ld = 1 if vector else max(2, m + (m & 1))   # vector: ONE right-hand side held contiguously (GEMV updates)
        R = self._buffer("rhs%d" % ld, self.grid.nlr(self.p) * self.nb, ld)
        if zero:
            R.zero_()
        return R

    # right-hand sides from which a tile solve is a multiply by the tile's explicit inverse
    MANY_RHS = 512

    def _tile_inverse(self, k):
        """W_kk = L_kk^-1 of an owned diagonal tile, formed on first use after a factorisation and kept
        (nb^2 doubles per owned diagonal tile: 1 / T of the matrix in all).  With it a many-right-hand-side
        tile solve is ONE triangular-aware MFMA GEMM instead of a recursion over the tile's n_b / 128
        leaves (2048-tile, 4096 right-hand sides: 0.6 ms -> 0.26 ms; the gradient's 1024 tile solves at
        n = 65536 cost 0.3 s more without it)."""
        W = self._winv.get(k)
        if W is None:
            hk = self.grid.tile_h(k)
            store = self.__dict__.setdefault("_winv_store", {})     # buffers outlive a factorisation, contents do not
            W = store.get(k)
            if W is None or W.shape[0] != hk:
                W = store[k] = self.ops.empty(hk, max(32, ((hk + 31) // 32) * 32))
            W[:, :hk].copy_(self.slabs[k][:hk, :hk])
            self.ops.trtri(W[:, :hk], self.dinv[k])
            self._winv[k] = W
        return W

    def _solve_tile(self, trans, k, Rk, m, ncols=None):
        hk = self.grid.tile_h(k)
        if m == 1:
            self.ops.trsm_left(trans, self.slabs[k][:hk, :hk], self.dinv[k], Rk[:, :1])
            return
        nc = Rk.shape[1] if ncols is None else ncols
        if nc >= self.MANY_RHS and hk > _lib.GPX_NB:
            tmp = self._buffer("tile_tmp%d" % Rk.shape[1], self.nb, Rk.shape[1])[:hk, :nc]
            self.ops.trmm(trans, self._tile_inverse(k)[:, :hk], Rk[:, :nc], tmp)
            Rk[:, :nc].copy_(tmp)
        else:
            self.ops.trsm_left(trans, self.slabs[k][:hk, :hk], self.dinv[k], Rk[:, :nc])

    def _live_cols(self, k, ld, tri):
        """Columns of tile row k that a slab solve has to touch.  tri = (t0, t1): the right-hand side is
        the identity on tile rows [t0, t1) (a slab of columns of Ky^-1), so in tile row k < t1 only the
        first (k - t0 + 1) nb columns are non-zero on the way forward, and only those are wanted (on or
        below the diagonal) on the way back."""
        if tri is None or k >= tri[1]:
            return ld
        return min(ld, (k - tri[0] + 1) * self.nb)

    def solve_forward(self, R, m, t0=0, tri=None):
        """R <- L^-1 R (regression.py:37-38,50) on tile rows >= t0 (the right-hand side must be zero above
        t0).  Right-looking, fan-in: after X[k] is known the ranks of process column k mod Q subtract
        L[I, k] X[k] from the partial sums of their rows; the partial sums of row k are reduced along
        process row k mod P onto the diagonal owner, which solves and broadcasts X[k] down its column."""
        g, ops, nb, P, Q, p, q = self.grid, self.ops, self.nb, self.P, self.Q, self.p, self.q
        ld = R.shape[1]
        xb = self._buffer("xb%d" % ld, nb, ld)
        for k in range(t0, g.T):
            pk, qk, hk = k % P, k % Q, g.tile_h(k)
            Rk = R[g.lrow(k): g.lrow(k) + hk]
            nc = self._live_cols(k, ld, tri)
            if p == pk:
                if self.rc:
                    self._reduce(Rk, self._grank(pk, qk), self.row_group)
                if q == qk:
                    self._solve_tile(0, k, Rk, m, nc)
            if q == qk and k < g.T - 1:
                Xk = Rk if p == pk else xb[:hk]
                if self.cc:
                    self._bcast(Xk, self._grank(pk, qk), self.col_group)
                rows = g.rows_from(p, k + 1)
                if rows > 0:
                    I1 = g.first_at_or_after(k + 1, p, P)
                    start = nb if p == pk else 0
                    Lk, tail = self.slabs[k][start: start + rows, :hk], R[g.lrow(I1): g.lrow(I1) + rows]
                    if ld == 1:
                        ops.gemv_acc(0, -1.0, Lk, Xk.view(-1), 1.0, tail.view(-1))
                    else:
                        ops.gemm(0, 0, rows, nc, hk, -1.0, Lk, Xk, 1.0, tail)

    def solve_backward(self, R, m, t0=0, tri=None):
        """R <- L^-T R (the second half of cho_solve behind regression.py:61) on tile rows >= t0.
        Left-looking, fan-in, the mirror image of solve_forward: the ranks of process column k mod Q
        form L[I>k, k]^T X[I>k] from the solution tiles resident in their rows, the partial results are
        reduced down the column onto the diagonal owner, which solves X[k] and broadcasts it along its
        process row (every rank of the row keeps it for the steps below)."""
        g, ops, nb, P, Q, p, q = self.grid, self.ops, self.nb, self.P, self.Q, self.p, self.q
        ld = R.shape[1]
        part = self._buffer("part%d" % ld, nb, ld)
        for k in range(g.T - 1, t0 - 1, -1):
            pk, qk, hk = k % P, k % Q, g.tile_h(k)
            Rk = R[g.lrow(k): g.lrow(k) + hk]
            nc = self._live_cols(k, ld, tri)
            if q == qk:
                rows = g.rows_from(p, k + 1)
                mine = p == pk
                tgt = Rk if mine else part[:hk]
                if nc < ld and not mine:
                    tgt.zero_()          # the columns the partial product skips still travel in the reduce
                if rows > 0:
                    I1 = g.first_at_or_after(k + 1, p, P)
                    start = nb if mine else 0
                    Lk, below = self.slabs[k][start: start + rows, :hk], R[g.lrow(I1): g.lrow(I1) + rows]
                    if ld == 1:
                        ops.gemv_acc(1, -1.0, Lk, below.view(-1), 1.0 if mine else 0.0, tgt.view(-1))
                    else:
                        ops.gemm(1, 0, hk, nc, rows, -1.0, Lk, below, 1.0 if mine else 0.0, tgt)
                elif not mine:
                    tgt.zero_()
                if self.cc:
                    self._reduce(tgt, self._grank(pk, qk), self.col_group)
                if mine:
                    self._solve_tile(1, k, Rk, m, nc)
            if p == pk and self.rc:
                self._bcast(Rk, self._grank(pk, qk), self.row_group)

    def _own_tiles(self, t0=0):
        return self.grid.diag_tiles(self.p, self.q, t0)

    def _gather_vector(self, R):
        """Column 0 of the solution container -> the full n-vector on every rank (all-reduce of n doubles)."""
        g, nb = self.grid, self.nb
        v = self.ops.zeros(max(self.n, 2))
        for k in self._own_tiles():
            hk = g.tile_h(k)
            v[k * nb: k * nb + hk].copy_(R[g.lrow(k): g.lrow(k) + hk, 0])
        return self._all_reduce(v)

    def solve_lml(self):
        """a = L^-1 y, alpha = L^-T a, LML (regression.py:50-55, 61); every rank ends with the full a and
        alpha (n doubles each) and the same LML."""
        g, ops, t, n, nb = self.grid, self.ops, self.torch, self.n, self.nb
        _lib.check(self.info_value, "distributed potrf")
        R = self.rhs_buffer(1, vector=True)
        for k in self._own_tiles():
            hk = g.tile_h(k)
            R[g.lrow(k): g.lrow(k) + hk, 0].copy_(self.y[k * nb: k * nb + hk])
        self.solve_forward(R, 1)
        self.a = self._gather_vector(R)
        logdet = 0.0
        for k in self._own_tiles():
            hk = g.tile_h(k)
            logdet += ops.logdet(self.slabs[k][:hk, :hk])
        self.solve_backward(R, 1)
        self.alpha = self._gather_vector(R)
        aa = ops.dot(n, self.a, self.a)
        ld_all = self._all_reduce(t.tensor([logdet], dtype=t.float64, device=self.store.device))
        self.lml = -0.5 * aa - float(ld_all.item()) - 0.5 * n * np.log(2.0 * np.pi)
        return self.lml

    # ------------------------------------------------------------------ gradient
    def grad(self):
        """regression.py:57-76 on the sharded factor: Ky^-1 is formed by slabs of `grad_block` columns,
        Kinv[c0:, slab] = L^-T (L^-1 [0; I; 0]) (one forward and one backward distributed solve on the
        tile rows >= the slab's first; 2 n^3 / 3 flops in all, as trtri + lauum), and every diagonal owner
        contracts the tile rows it ends up with against dK/dtheta on the fly (gpx_grad_trace_slab) --
        Ky^-1 never exists whole.  One all-reduce of d + 2 doubles.
        Returns [d/d noise, d/d cov sigma, d/d cov beta...] (the reference's key order)."""
        g, ops, t, n, nb = self.grid, self.ops, self.torch, self.n, self.nb
        nbeta = len(self.beta)
        total = np.zeros(2 + nbeta)
        tiles_per_slab = max(1, self.grad_block // nb)
        for t0 in range(0, g.T, tiles_per_slab):
            t1 = min(g.T, t0 + tiles_per_slab)
            w = sum(g.tile_h(J) for J in range(t0, t1))
            R = self.rhs_buffer(w)
            for k in self._own_tiles(t0):
                if k < t1:
                    hk = g.tile_h(k)
                    R[g.lrow(k): g.lrow(k) + hk, (k - t0) * nb: (k - t0) * nb + hk].diagonal().fill_(1.0)
            self.solve_forward(R, w, t0, tri=(t0, t1))
            self.solve_backward(R, w, t0, tri=(t0, t1))
            for k in self._own_tiles(t0):
                hk = g.tile_h(k)
                total += ops.grad_trace_slab(self.kind, self.sigma, self.beta, self.X, self.d, self.alpha,
                                             R[g.lrow(k): g.lrow(k) + hk, :w], k * nb, t0 * nb)
        tot = t.from_numpy(total).to(self.store.device)
        tot = self._all_reduce(tot)
        return tot.cpu().numpy()

    # ------------------------------------------------------------------ predict
    def predict(self, Xs, full_cov=True):
        """regression.py:31-44.  Returns (mean, cov) -- or (mean, diagonal variance) with
        full_cov=False, which never forms the m x m matrix -- as host arrays on every rank."""
        mean, second = self.predict_device(self.upload_test(Xs), full_cov=full_cov)
        return mean.cpu().numpy(), second.cpu().numpy()

    def predict_device(self, Xs_d, full_cov=True):
        """V = L^-1 k* by the distributed forward solve (the tile rows of V end on the diagonal owners),
        mean = V^T a and cov = K** - V^T V + noise I (or its diagonal) from each owner's rows, all-reduced."""
        g, ops, t, n, d, nb = self.grid, self.ops, self.torch, self.n, self.d, self.nb
        m = Xs_d.shape[0]
        R = self.rhs_buffer(m)
        ld = R.shape[1]
        own = self._own_tiles()
        for k in own:
            hk = g.tile_h(k)
            ops.gram(self.kind, self.sigma, self.beta, self.X[k * nb: k * nb + hk, :d], Xs_d[:, :d],
                     R[g.lrow(k): g.lrow(k) + hk, :m], 0.0)
        self.solve_forward(R, m)
        # this process's rows of V and of a, compacted (plumbing copies; the diagonal owners' tile rows
        # are every (L/P)-th local tile row)
        rows = max(len(own) * nb, 2)
        Vc = self._buffer("Vc%d" % ld, rows, ld)
        ac = self._buffer("ac", rows, 1).view(-1)
        Vc.zero_()
        ac.zero_()
        for i, k in enumerate(own):
            hk = g.tile_h(k)
            Vc[i * nb: i * nb + hk].copy_(R[g.lrow(k): g.lrow(k) + hk])
            ac[i * nb: i * nb + hk].copy_(self.a[k * nb: k * nb + hk])
        Vc = Vc[:rows]
        mean = ops.zeros(max(m, 2))
        ops.gemv_t(Vc[:, :m], ac, mean)
        mean = self._all_reduce(mean)[:m]
        if not full_cov:
            var = ops.zeros(max(m, 2))
            ops.colsumsq(Vc[:, :m], (self.sigma + self.noise) if self.rank == 0 else 0.0, var)
            return mean, self._all_reduce(var)[:m]
        ldc = max(32, ((m + 31) // 32) * 32)
        cov = ops.zeros(m, ldc)
        if self.rank == 0:
            ops.gram(self.kind, self.sigma, self.beta, Xs_d[:, :d], Xs_d[:, :d], cov[:, :m], self.noise)
        ops.gemm(1, 0, m, m, rows, -1.0, Vc, Vc, 1.0, cov)
        return mean, self._all_reduce(cov)[:, :m]

    # ------------------------------------------------------------------ checks / diagnostics
    def gather_lower(self):
        """The whole lower-triangular matrix / factor on every rank as an n x n host array (strict
        upper part zero).  Diagnostics and tests only: it undoes the sharding."""
        g, nb, n = self.grid, self.nb, self.n
        full = self.torch.zeros(n, n, dtype=self.torch.float64)
        for J, slab in self.slabs.items():
            wJ = g.tile_h(J)
            host = slab.cpu()
            for i, I in enumerate(g.local_rows(self.p, J)):
                hI = g.tile_h(I)
                blk = host[i * nb: i * nb + hI, :wJ]
                full[I * nb: I * nb + hI, J * nb: J * nb + wJ] = blk.tril() if I == J else blk
        return self._all_reduce_host(full).numpy()

    def gather_dinv(self):
        """All 128 x 128 diagonal-block inverses (ceil(n/128) x 128 x 128) on every rank; tests only."""
        g, nb, n = self.grid, self.nb, self.n
        nblk = (n + 127) // 128
        full = self.torch.zeros(nblk, 128, 128, dtype=self.torch.float64)
        for k, dk in self.dinv.items():
            cnt = dk.numel() // (128 * 128)
            full[k * (nb // 128): k * (nb // 128) + cnt] = dk.cpu().view(cnt, 128, 128)
        return self._all_reduce_host(full).numpy()

    def _all_reduce_host(self, full):
        if self.world == 1:
            return full
        if self._gloo:
            self.dist.all_reduce(full, group=self.group)
            return full
        dev = full.to(self.store.device)
        self.dist.all_reduce(dev, group=self.group)
        return dev.cpu()

    def residual(self, chunk=2048):
        """|| Ky alpha - y || / || y || with Ky rebuilt row chunk by row chunk into a scratch buffer, the
        chunks dealt round-robin to the ranks (a size-independent parity property for the full-size
        runs; the factor is not touched)."""
        ops, t, n, d = self.ops, self.torch, self.n, self.d
        ldn = max(32, ((n + 31) // 32) * 32)
        scratch = self._buffer("resid", min(chunk, n), ldn)
        out = ops.zeros(max(min(chunk, n), 2))
        part = t.zeros(2, dtype=t.float64, device=self.store.device)
        for i, r0 in enumerate(range(0, n, chunk)):
            if i % self.world != self.rank:
                continue
            r1 = min(n, r0 + chunk)
            slab = scratch[: r1 - r0, :n]
            ops.gram(self.kind, self.sigma, self.beta, self.X[r0:r1, :d], self.X[:, :d], slab, 0.0)
            ops.gemv_n(slab, self.alpha, out)
            res = out[: r1 - r0] + self.noise * self.alpha[r0:r1] - self.y[r0:r1]
            part[0] += (res * res).sum()
            part[1] += (self.y[r0:r1] ** 2).sum()
        part = self._all_reduce(part)
        return float((part[0] / part[1]).sqrt().item())
