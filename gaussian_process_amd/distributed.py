"""Multi-GPU exact GP regression: block-column-cyclic Cholesky with panel broadcasts over
``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

One process per GPU.  The reference has no distributed code at all; what is distributed here is
the arithmetic of regression.py:31-55 (Gram, ``cholesky``, the triangular solves, LML, predictive
mean / covariance), with the same kernels as the single-GPU path (libgpx) doing all local work.

Layout and algorithm (realised 1-D for round 1):
  * every rank holds an n x n row-major buffer; block column j (width ``block``, rows >= j*block)
    is OWNED by rank j % P, which builds it (Gram, no communication: X is replicated) and applies
    every trailing update to it;
  * step k: the owner factors the diagonal block (``gpx_potrf_async``), solves the panel below it
    (``gpx_trsm_right``), packs panel k (+ its 128x128 diagonal-block inverses) into a contiguous
    staging buffer and broadcasts it in row pieces; every rank files the panel into its buffer, so
    after the last step EVERY rank holds the complete factor L, and applies the filed panels to its
    own block columns j > k a few at a time, ``A[j0:, j] -= A[j0:, c0:c1] A[j, c0:c1]^T`` (one MFMA
    GEMM with a long K instead of one per panel);
  * look-ahead: the owner of panel k+1 brings its column up to date with panel k piece by piece as
    the pieces arrive, factors it and posts its own pieces (async, RCCL's own stream) before the bulk
    updates of step k are enqueued, so the transfers and the panel factorisation hide under the
    other ranks' GEMMs;
  * solves after the factorisation are split by right-hand-side column: rank r computes
    ``V[:, cols_r] = L^-1 k*(:, cols_r)``; V is all-gathered for the ``V^T V`` block rows.
Communication per step: the broadcasts of one panel, (n - k*block) x block doubles in all (the whole
lower triangle once per rank over the factorisation, 17 GB at n = 65536) -- no other data-path collective.
"""
import ctypes as C

import numpy as np

from . import _lib


def owner_of(block_index, world):
    return block_index % world


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

    def trsm_left(self, trans, L, dinv, B):
        _lib.check(self.lib.gpx_trsm(self.h, trans, L.shape[0], B.shape[1], self._p(L), L.stride(0),
                                     self._p(dinv), self._p(B), B.stride(0)), "gpx_trsm")

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

    def grad_trace_slab(self, kind, sigma, beta, X, d, alpha, slab, row_off, col_off):
        barr, bp = _lib.host_doubles(beta)
        out = (C.c_double * (2 + len(barr)))()
        _lib.check(self.lib.gpx_grad_trace_slab(self.h, kind, self._p(X), X.shape[0], X.stride(0), d, float(sigma),
                                                bp, self._p(alpha), self._p(slab), slab.stride(0), int(row_off),
                                                int(col_off), slab.shape[0], slab.shape[1],
                                                C.cast(out, C.POINTER(C.c_double))), "gpx_grad_trace_slab")
        return np.array(list(out))

    # ---- a second stream for work that should overlap the main stream (the diagonal-block potrf of
    # the look-ahead panel is a chain of one-workgroup kernels; the column update fills the other CUs)
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

    def sync(self):
        self.torch.cuda.synchronize(self.device)


class DistributedGP:
    """Exact GP regression over the ranks of a torch.distributed process group."""

    def __init__(self, ops, dist, rank, world, block=2048, group=None, update_k=None, grad_block=None,
                 row_chunk=None):
        assert block % _lib.GPX_NB == 0, "block must be a multiple of 128"
        self.ops, self.dist, self.rank, self.world, self.block, self.group = ops, dist, rank, world, block, group
        self.torch = ops.torch
        # K the delayed trailing updates aim for (see potrf).  Longer K is cheaper per flop (world 1,
        # block 512: potrf 1.77 s at K = 1024, 1.67 at 2048, 1.63 at 4096, 1.62 at 8192) but one such GEMM
        # must stay short next to a factorisation step, because the next panel's chain queues behind
        # it on the same stream: 8192 on one rank, 4096 on two, 2048 beyond.
        self.update_k = update_k or {1: 8192, 2: 4096}.get(world, 2048)
        # Rows per piece of a panel broadcast (absolute row grid, a multiple of the block width): the
        # owner of the next panel starts on a piece as soon as it has arrived and sends its own piece
        # on at once, so the factorisation chain carries one piece per step, not a whole panel.
        if row_chunk is None:
            row_chunk = max(block, 8192 if world >= 4 else 16384)
        assert row_chunk % block == 0, "row_chunk must be a multiple of block"
        self.row_chunk = row_chunk
        # Column width of the gradient's own partition.  Wide slabs cut the number of launches (every
        # slab solve walks all 128-row leaves of L: measured at world 1, n = 65536: 4.78 s at 1024,
        # 3.75 s at 2048, 3.33 s at 4096), narrow ones balance the ranks (max/mean load of the snake
        # order at 8 ranks: 1.03 at 1024, 1.10 at 2048, 1.37 at 4096).
        self.grad_block = grad_block or (4096 if world <= 4 else 2048)
        self._gloo = dist.get_backend(group) == "gloo"
        self.n = 0

    # ------------------------------------------------------------------ helpers
    def _blocks(self):
        b = self.block
        return [(k, k * b, min(b, self.n - k * b)) for k in range((self.n + b - 1) // b)]

    def _dinv_slice(self, k0, kb):
        nb = _lib.GPX_NB
        lo = (k0 // nb) * nb * nb
        return self.dinv[lo: lo + ((kb + nb - 1) // nb) * nb * nb]

    def _bcast(self, t, src, async_op=False):
        """Broadcast a contiguous device buffer.  RCCL: asynchronous on RCCL's stream, ordered after
        the kernels already enqueued; wait() makes the compute stream wait, not the host.  gloo
        (rehearsals with several ranks on one GPU, CPU tests): staged through the host and
        completed before returning."""
        if t.is_cuda and self._gloo:
            host = t.cpu()                       # synchronises with the kernels that produced t
            self.dist.broadcast(host, src=src, group=self.group)
            if self.rank != src:
                t.copy_(host)
            return _Done()
        w = self.dist.broadcast(t, src=src, group=self.group, async_op=async_op)
        return w if async_op else _Done()

    def _all_reduce(self, t, op):
        if t.is_cuda and self._gloo:
            host = t.cpu()
            self.dist.all_reduce(host, op=op, group=self.group)
            return host
        self.dist.all_reduce(t, op=op, group=self.group)
        return t

    def _all_gather(self, outs, mine):
        """all_gather; gloo has no device all_gather, so there it is P broadcasts (test rigs only)."""
        if mine.is_cuda and self._gloo:
            for s in range(self.world):
                if s == self.rank:
                    outs[s].copy_(mine)
                self._bcast(outs[s], s)
        else:
            self.dist.all_gather(outs, mine, group=self.group)

    # ------------------------------------------------------------------ fit
    def allocate(self, X, y, kind, sigma, beta, noise):
        """Upload the (replicated) inputs and allocate everything fit/predict keep resident."""
        ops, t = self.ops, self.torch
        n, d = X.shape
        self.n, self.d = n, d
        self.set_theta(kind, sigma, beta, noise)
        ld = max(32, ((n + 31) // 32) * 32)
        self.A = ops.empty(n, ld)
        self.X = ops.zeros(n, max(32, ((d + 31) // 32) * 32))
        self.X[:, :d].copy_(t.from_numpy(np.ascontiguousarray(X, dtype=np.float64)))
        self.y = ops.empty(max(n, 2))
        self.y[:n].copy_(t.from_numpy(np.ascontiguousarray(y, dtype=np.float64)))
        self.dinv = ops.empty(max(((n + 127) // 128) * 128 * 128, 2))
        self.info = ops.zeros(64, dtype=t.int32)
        b = min(self.block, ((n + 127) // 128) * 128)
        cap = n * b + ((b + 127) // 128) * 128 * 128
        self.stg = [ops.empty(cap), ops.empty(cap)]     # double-buffered panel staging
        self.rhs = ops.zeros(n, 2)

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
        every rank.  Returns the LML; the factor stays resident on every rank for predict()."""
        self.allocate(X, y, kind, sigma, beta, noise)
        self.build_gram()
        self.potrf()
        return self.solve_lml()

    def build_gram(self):
        """Each rank fills only the block columns it owns (rows from the diagonal block down)."""
        for k, k0, kb in self._blocks():
            if owner_of(k, self.world) != self.rank:
                continue
            Xv = self.X[:, : self.d]
            self.ops.gram(self.kind, self.sigma, self.beta, Xv[k0:], Xv[k0:k0 + kb],
                          self.A[k0:self.n, k0:k0 + kb], self.noise)

    def _row_chunks(self, k0):
        """Row ranges [r0, r1) of panel k (rows k0..n) on the absolute row_chunk grid."""
        n, rc = self.n, self.row_chunk
        cuts = [k0] + list(range((k0 // rc + 1) * rc, n, rc)) + [n]
        return [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]

    def _panel_len(self, k0, kb):
        return (self.n - k0) * kb + self._dinv_slice(k0, kb).numel()

    def _send_panel(self, k, k0, kb, owner, produce=None):
        """Broadcast panel k (in stg[k % 2]) piece by piece, the diagonal-block inverses last; every
        rank issues the same sequence of collectives.  On the owner `produce(r0, r1)` computes and
        packs the rows [r0, r1) right before they are sent.  Returns [(last row + 1, handle), ...]."""
        stg = self.stg[k % 2]
        nk = self.n - k0
        handles = []
        for r0, r1 in self._row_chunks(k0):
            if produce is not None:
                produce(r0, r1)
            handles.append((r1, self._bcast(stg[(r0 - k0) * kb:(r1 - k0) * kb], owner, async_op=True)))
        dlen = self._dinv_slice(k0, kb).numel()
        handles.append((None, self._bcast(stg[nk * kb: nk * kb + dlen], owner, async_op=True)))
        return handles

    def _producer(self, k, k0, kb, prev=None):
        """Owner side of panel k as a per-piece callback for _send_panel.  prev = (Pprev, kprev0,
        handles of panel k-1) or None for the first panel.  First piece: (look-ahead: bring the
        diagonal block up to date with panel k-1,) potrf of the diagonal block -- on the side stream
        while the main stream updates the rows below it --, then per piece: update with panel k-1,
        TRSM against the diagonal block, pack.  A piece only waits for the piece of panel k-1 that
        holds the same rows."""
        n, ops = self.n, self.ops
        stg = self.stg[k % 2]
        Akk = self.A[k0:k0 + kb, k0:k0 + kb]
        dk = self._dinv_slice(k0, kb)
        state = {"arrived": 0}

        def wait_rows(r1):
            """Make the main stream wait for the pieces of panel k-1 up to the one holding row r1 - 1."""
            if prev is None:
                return
            hs, i = prev[2], state["arrived"]
            while i < len(hs) and hs[i][0] is not None and (i == 0 or hs[i - 1][0] < r1):
                hs[i][1].wait()
                i += 1
            state["arrived"] = i

        def produce(r0, r1):
            wait_rows(r1)
            first = r0 == k0
            lo = k0 + kb if first else r0
            token = None
            if prev is not None:
                Pp, kp0 = prev[0], prev[1]
                top = Pp[k0 - kp0: k0 - kp0 + kb]
                if first:
                    ops.gemm(0, 1, kb, kb, Pp.shape[1], -1.0, top, top, 1.0, Akk)
                    ops.side_begin()
                    ops.potrf_async(Akk, dk, self.info, k0)
                    token = ops.side_end()
                if r1 > lo:
                    ops.gemm(0, 1, r1 - lo, kb, Pp.shape[1], -1.0, Pp[lo - kp0: r1 - kp0], top, 1.0,
                             self.A[lo:r1, k0:k0 + kb])
                if token is not None:
                    ops.side_join(token)
            elif first:
                ops.potrf_async(Akk, dk, self.info, k0)
            if r1 > lo:
                ops.trsm_right(self.A[lo:r1, k0:k0 + kb], Akk, dk)
            stg[(r0 - k0) * kb:(r1 - k0) * kb].view(r1 - r0, kb).copy_(self.A[r0:r1, k0:k0 + kb])
            if first:
                nk = n - k0
                stg[nk * kb: nk * kb + dk.numel()].copy_(dk)

        return produce

    def _apply_filed(self, j0, jb, c0, c1):
        """A[j0:, J] -= A[j0:, c0:c1] A[J, c0:c1]^T: the panels filed in columns [c0, c1) applied to
        block column J = [j0, j0+jb) in ONE GEMM with K = c1 - c0."""
        if c1 <= c0:
            return
        n = self.n
        self.ops.gemm(0, 1, n - j0, jb, c1 - c0, -1.0, self.A[j0:n, c0:c1], self.A[j0:j0 + jb, c0:c1], 1.0,
                      self.A[j0:n, j0:j0 + jb])

    def potrf(self):
        """Distributed right-looking Cholesky with DELAYED trailing updates and a row-pipelined
        panel chain.  Every rank files every panel into its own buffer, so the panels a block column
        still has to absorb sit side by side in A and are applied together: one GEMM with
        K = (number of pending panels) x block instead of one short-K GEMM per panel (K = block is
        15-20 % slower on the MFMA kernel and rewrites the column every step).  A column is forced up
        to date one step before it becomes the look-ahead column, so the critical chain only ever
        applies the newest panel -- and it does so piece by piece (row_chunk rows): the owner of panel
        k+1 updates, solves and sends on the rows of a piece as soon as that piece of panel k is in,
        so a panel's broadcast overlaps the next panel's factorisation instead of preceding it."""
        ops, t, n, b = self.ops, self.torch, self.n, self.block
        blocks = self._blocks()
        self.info.zero_()
        group = max(1, self.update_k // b)                # panels per delayed update
        applied = {j: 0 for j, _, _ in blocks}            # first panel column NOT yet applied to column j
        k, k0, kb = blocks[0]
        o0 = owner_of(0, self.world)
        handles = self._send_panel(0, k0, kb, o0, self._producer(0, k0, kb) if o0 == self.rank else None)
        for k, k0, kb in blocks:
            cur = self.stg[k % 2]
            nk = n - k0
            Pk = cur[: nk * kb].view(nk, kb)
            mine = owner_of(k, self.world) == self.rank
            nxt = None
            if k + 1 < len(blocks):
                k1, k10, k1b = blocks[k + 1]
                o1 = owner_of(k1, self.world)
                produce = None
                if o1 == self.rank:  # look-ahead: bring panel k+1 up to date, factor it, send it
                    self._apply_filed(k10, k1b, applied[k1], k0)      # older panels (normally none)
                    produce = self._producer(k1, k10, k1b, (Pk, k0, handles))
                    applied[k1] = k0 + kb
                nxt = self._send_panel(k1, k10, k1b, o1, produce)
            for _, h in handles:
                h.wait()
            if not mine:  # file panel k (and its diagonal inverses) so every rank ends with all of L
                self.A[k0:n, k0:k0 + kb].copy_(Pk)
                dk = self._dinv_slice(k0, kb)
                dk.copy_(cur[nk * kb: nk * kb + dk.numel()])
            # The rank that factors panel k+2 next step enqueues only that column's catch-up now: its
            # other updates would sit in the stream ahead of the chain it is about to run, and lose
            # nothing by waiting one step (they come back with a longer K).
            chain_next = k + 2 < len(blocks) and owner_of(k + 2, self.world) == self.rank
            for j, j0, jb in blocks[k + 2:]:
                if owner_of(j, self.world) != self.rank:
                    continue
                pending = (k0 + kb - applied[j] + b - 1) // b
                if j == k + 2 or (pending >= group and not chain_next):
                    self._apply_filed(j0, jb, applied[j], k0 + kb)
                    applied[j] = k0 + kb
            handles = nxt
        # LAPACK info: smallest failing leading minor over all ranks (0 = success)
        big = 2 ** 30
        loc = int(self.info[0].item())
        val = t.tensor([loc if loc > 0 else big], dtype=t.int64, device=self.A.device)
        val = self._all_reduce(val, self.dist.ReduceOp.MIN)
        self.info_value = 0 if int(val.item()) >= big else int(val.item())
        return self.info_value

    def solve_lml(self):
        """a = L^-1 y, alpha = L^-T a, LML (regression.py:50-55, 61); replicated on every rank."""
        ops, t, n = self.ops, self.torch, self.n
        _lib.check(self.info_value, "distributed potrf")
        L = self.A[:n, :n]
        rhs = self.rhs
        rhs[:, 0].copy_(self.y[:n])
        ops.trsm_left(0, L, self.dinv, rhs[:, :1])
        self.a = rhs[:, 0].clone()
        ops.trsm_left(1, L, self.dinv, rhs[:, :1])
        self.alpha = rhs[:, 0].clone()
        aa = ops.dot(n, self.a, self.a)
        self.lml = -0.5 * aa - ops.logdet(L) - 0.5 * n * np.log(2.0 * np.pi)
        return self.lml

    # ------------------------------------------------------------------ gradient
    def grad(self):
        """regression.py:57-76 without ever forming Ky^-1 whole and without communication: every rank
        holds all of L, so the owner of block column J computes the part of Ky^-1 on and below the
        diagonal of J from the trailing sub-factor alone,
            Kinv[j0:, J] = L[j0:, j0:]^-T ( L[j0:, j0:]^-1 [I; 0] )      ((n - j0)^2 b flops each, 2n^3/3 in all),
        and adds its share of the trace sums (gpx_grad_trace_slab).  One all-reduce of d + 2 doubles.
        Returns [d/d noise, d/d cov sigma, d/d cov beta...] (the reference's key order)."""
        ops, t, n = self.ops, self.torch, self.n
        nbeta = len(self.beta)
        total = np.zeros(2 + nbeta)
        # The gradient has its own column partition (every rank holds all of L, so any partition
        # works): wide slabs for GEMM efficiency.  The cost of a slab falls like (n - j0)^2, so plain
        # cyclic dealing would load rank 0 twice as heavily as the last rank; slabs are handed out
        # heaviest first to the least loaded rank instead (same deterministic list on every rank:
        # max/mean load 1.00 / 1.01 / 1.02 on 2 / 4 / 8 ranks at n = 65536, against 1.01 / 1.08 / 1.10
        # for dealing them in snake order r, 2P-1-r, ...).
        gb = min(self.grad_block, ((n + 127) // 128) * 128)
        if getattr(self, "slab", None) is None or self.slab.numel() < n * gb:
            self.slab = ops.empty(n * gb)
        nblk = (n + gb - 1) // gb
        loads, mine = [0] * self.world, []
        for k in range(nblk):            # k ascending = cost descending
            r = loads.index(min(loads))
            loads[r] += (n - k * gb) ** 2 * min(gb, n - k * gb)
            if r == self.rank:
                mine.append(k)
        for k in mine:
            k0, kb = k * gb, min(gb, n - k * gb)
            nk, lds = n - k0, kb + (kb & 1)
            S = self.slab[: nk * lds].view(nk, lds)
            S.zero_()
            S[:kb, :kb].diagonal().fill_(1.0)
            S = S[:, :kb]
            Lsub = self.A[k0:n, k0:n]
            dsub = self.dinv[(k0 // _lib.GPX_NB) * _lib.GPX_NB * _lib.GPX_NB:]
            ops.trsm_left(0, Lsub, dsub, S)
            ops.trsm_left(1, Lsub, dsub, S)
            total += ops.grad_trace_slab(self.kind, self.sigma, self.beta, self.X, self.d, self.alpha, S, k0, k0)
        tot = t.from_numpy(total).to(self.A.device)
        tot = self._all_reduce(tot, self.dist.ReduceOp.SUM)
        return tot.cpu().numpy()

    # ------------------------------------------------------------------ predict
    def predict(self, Xs):
        """regression.py:31-44 with the m test points split over the ranks.  Returns (mean, cov) as
        host arrays on every rank."""
        mean, cov = self.predict_device(self.upload_test(Xs))
        return mean.cpu().numpy(), cov.cpu().numpy()

    def predict_device(self, Xs_d):
        ops, t, n, d, P, r = self.ops, self.torch, self.n, self.d, self.world, self.rank
        m = Xs_d.shape[0]
        mc = (m + P - 1) // P
        mc = max(2, mc + (mc & 1))                      # even chunk width (16-byte rows)
        c0, c1 = min(m, r * mc), min(m, (r + 1) * mc)
        L = self.A[:n, :n]
        V = ops.zeros(n, mc)
        if c1 > c0:
            ops.gram(self.kind, self.sigma, self.beta, self.X[:, :d], Xs_d[c0:c1, :d], V[:, : c1 - c0], 0.0)
            ops.trsm_left(0, L, self.dinv, V[:, : c1 - c0])
        mean_loc = ops.zeros(mc)
        ops.gemv_t(V, self.a, mean_loc)
        means = [ops.empty(mc) for _ in range(P)]
        self._all_gather(means, mean_loc)
        mean = t.cat([means[s][: max(0, min(m, (s + 1) * mc) - min(m, s * mc))] for s in range(P)])
        Vs = [ops.empty(n, mc) for _ in range(P)]
        self._all_gather(Vs, V)
        mpad = P * mc
        cov_loc = ops.zeros(mc, mpad + (mpad & 1))
        if c1 > c0:
            for s in range(P):
                s0, s1 = min(m, s * mc), min(m, (s + 1) * mc)
                if s1 <= s0:
                    continue
                blk = cov_loc[: c1 - c0, s * mc: s * mc + (s1 - s0)]
                ops.gram(self.kind, self.sigma, self.beta, Xs_d[c0:c1, :d], Xs_d[s0:s1, :d], blk,
                         self.noise if s == r else 0.0)
                ops.gemm(1, 0, c1 - c0, s1 - s0, n, -1.0, V, Vs[s], 1.0, blk)
        covs = [ops.empty(mc, cov_loc.shape[1]) for _ in range(P)]
        self._all_gather(covs, cov_loc)
        rows = []
        for s in range(P):
            s0, s1 = min(m, s * mc), min(m, (s + 1) * mc)
            if s1 > s0:
                rows.append(t.cat([covs[s][: s1 - s0, q * mc: q * mc + max(0, min(m, (q + 1) * mc) - min(m, q * mc))]
                                   for q in range(P)], dim=1))
        return mean, t.cat(rows, dim=0)

    # ------------------------------------------------------------------ checks
    def residual(self):
        """|| Ky alpha - y || / || y || with Ky rebuilt row slab by row slab (a size-independent
        parity property for the full-size runs; overwrites the factor)."""
        ops, t, n, d, P, r = self.ops, self.torch, self.n, self.d, self.world, self.rank
        rows = ((n + P - 1) // P + 1) // 2 * 2
        r0, r1 = min(n, r * rows), min(n, (r + 1) * rows)
        part = t.zeros(2, dtype=t.float64, device=self.A.device)
        if r1 > r0:
            slab = self.A[r0:r1, :n]
            ops.gram(self.kind, self.sigma, self.beta, self.X[r0:r1, :d], self.X[:, :d], slab, 0.0)
            out = ops.zeros(max(r1 - r0, 2))
            ops.gemv_n(slab, self.alpha, out)
            res = out[: r1 - r0] + self.noise * self.alpha[r0:r1] - self.y[r0:r1]
            part[0] = (res * res).sum()
            part[1] = (self.y[r0:r1] ** 2).sum()
        part = self._all_reduce(part, self.dist.ReduceOp.SUM)
        return float((part[0] / part[1]).sqrt().item())
