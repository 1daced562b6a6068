"""Test double for gaussian_process_amd.distributed.GpuOps: the same per-rank primitive
interface on CPU torch tensors with NumPy/SciPy arithmetic.  TEST INFRASTRUCTURE ONLY -- it lets
the world_size>1 control flow (ownership, look-ahead, staging, broadcasts, info reduction, the
RHS-split predict) run under gloo on a machine without a GPU.  Never imported by the package."""
import numpy as np
import scipy.linalg as sla
import torch

from oracle import gp_oracle as orc

KIND_NAME = {0: "rbf", 1: "ard"}
NB = 128


class NumpyOps:
    torch = torch
    device = torch.device("cpu")

    def empty(self, *shape, dtype=None):
        # poison instead of "uninitialised": reads of never-written memory show up as NaN
        t = torch.empty(shape, dtype=dtype or torch.float64)
        if t.dtype == torch.float64:
            t.fill_(float("nan"))
        return t

    def zeros(self, *shape, dtype=None):
        return torch.zeros(shape, dtype=dtype or torch.float64)

    def gram(self, kind, sigma, beta, Xr, Xc, out, diag_add):
        K = orc.cov(KIND_NAME[kind], Xr.numpy().copy(), Xc.numpy().copy(), sigma, beta)[0]
        k = min(K.shape)
        K[np.arange(k), np.arange(k)] += diag_add
        out.numpy()[...] = K

    def potrf_async(self, Akk, dinv, info, info_offset):
        a = Akk.numpy()
        n = a.shape[0]
        sym = np.tril(a) + np.tril(a, -1).T
        try:
            L = np.linalg.cholesky(sym)
        except np.linalg.LinAlgError:
            L = np.full_like(sym, np.nan)
            minor = next(i for i in range(1, n + 1) if not _is_pd(sym[:i, :i]))
            if int(info[0]) == 0:
                info[0] = info_offset + minor
        for b0 in range(0, n, NB):          # lower part + zeroed strict upper of diagonal blocks only
            b1 = min(n, b0 + NB)
            a[b0:b1, :b0] = L[b0:b1, :b0]
            a[b0:b1, b0:b1] = np.tril(L[b0:b1, b0:b1])
        self._dinv(L, dinv)

    @staticmethod
    def _dinv(L, dinv):
        n = L.shape[0]
        d = dinv.numpy()
        for i, b0 in enumerate(range(0, n, NB)):
            b1 = min(n, b0 + NB)
            blk = np.eye(NB)
            if np.isfinite(L[b0:b1, b0:b1]).all():
                blk[: b1 - b0, : b1 - b0] = sla.solve_triangular(L[b0:b1, b0:b1], np.eye(b1 - b0), lower=True)
            d[i * NB * NB:(i + 1) * NB * NB] = blk.reshape(-1)

    def trsm_right(self, Xp, Lkk, dinv):
        L = np.tril(Lkk.numpy())
        x = Xp.numpy()
        x[...] = sla.solve_triangular(L, x.T, lower=True, check_finite=False).T

    def gemm(self, ta, tb, M, N, K, alpha, A, B, beta, Cm):
        a = A.numpy()
        b = B.numpy()
        opa = a[:K, :M].T if ta else a[:M, :K]
        opb = b[:N, :K].T if tb else b[:K, :N]
        c = Cm.numpy()
        c[:M, :N] = alpha * (opa @ opb) + (beta * c[:M, :N] if beta != 0.0 else 0.0)

    def gemm_colblocks(self, M, N, K, alpha, A, B, beta, store, nb, row_base, blk_off, blk_row0):
        """Restatement of gpx_gemm_colblocks: column block c of the product goes to its own slab inside `store`,
        rows from blk_row0[c] on (128-row tiles that start above it are skipped)."""
        a, b, st = A.numpy()[:M, :K], B.numpy()[:N, :K], store.numpy()
        off, row0 = blk_off.numpy(), blk_row0.numpy()
        for c in range((N + nb - 1) // nb):
            w = min(nb, N - c * nb)
            diag_first = bool(int(row0[c]) & 1)                 # the slab starts with the column's diagonal tile
            r0c = int(row0[c]) & ~1
            r_first = max(0, r0c - row_base)                    # first live row of the launch for this block
            if r_first >= M:
                continue
            prod = a[r_first:] @ b[c * nb: c * nb + w].T
            for i in range(r_first, M):
                o = int(off[c]) + (row_base + i) * nb
                # inside the slab's diagonal tile the 128-tiles strictly above the diagonal are skipped
                rel = row_base + i - r0c
                w_live = min(w, (rel // NB + 1) * NB) if (diag_first and rel < nb) else w
                st[o: o + w_live] = alpha * prod[i - r_first][:w_live] + (beta * st[o: o + w_live] if beta != 0.0 else 0.0)

    def trsm_left(self, trans, L, dinv, B):
        l = np.tril(L.numpy())
        b = B.numpy()
        b[...] = sla.solve_triangular(l, b, lower=True, trans="T" if trans else "N", check_finite=False)

    def trtri(self, W, dinv):
        w = W.numpy()
        w[...] = np.tril(sla.solve_triangular(np.tril(w), np.eye(w.shape[0]), lower=True, check_finite=False))

    def trmm(self, trans, W, B, Out):
        w = np.tril(W.numpy())
        Out.numpy()[...] = (w.T if trans else w) @ B.numpy()

    def trmm_right(self, X, W, Out):
        Out.numpy()[...] = X.numpy() @ np.tril(W.numpy()).T

    def logdet(self, L):
        return float(np.log(np.diagonal(L.numpy())).sum())

    def dot(self, n, x, y):
        return float(x.numpy()[:n] @ y.numpy()[:n])

    def gemv_t(self, A, x, y):
        n, m = A.shape
        y.numpy()[:m] = A.numpy().T @ x.numpy()[:n]

    def gemv_n(self, A, x, y):
        n, m = A.shape
        y.numpy()[:n] = A.numpy() @ x.numpy()[:m]

    def gemv_acc(self, trans, alpha, A, x, beta, y):
        a = A.numpy()
        n, m = a.shape
        if trans:
            y.numpy()[:m] = alpha * (a.T @ x.numpy()[:n]) + (beta * y.numpy()[:m] if beta != 0.0 else 0.0)
        else:
            y.numpy()[:n] = alpha * (a @ x.numpy()[:m]) + (beta * y.numpy()[:n] if beta != 0.0 else 0.0)

    def colsumsq(self, V, base, var):
        m = V.shape[1]
        var.numpy()[:m] = base - (V.numpy() ** 2).sum(axis=0)

    def grad_trace_slab(self, kind, sigma, beta, X, d, alpha, slab, row_off, col_off):
        """Restatement of the slab sums: entries on/below the global diagonal, weight 2 off it."""
        Xn = X.numpy()[:, :d]
        al = alpha.numpy()
        S = slab.numpy()
        nr, nc = S.shape
        rows, cols = np.arange(row_off, row_off + nr), np.arange(col_off, col_off + nc)
        name = KIND_NAME[kind]
        Kb = orc.cov(name, Xn[rows], Xn[cols], sigma, beta)[0]
        A = np.outer(al[rows], al[cols]) - S
        w = np.where(rows[:, None] > cols[None, :], 2.0, np.where(rows[:, None] == cols[None, :], 1.0, 0.0))
        out = [0.5 * np.sum(np.where(rows[:, None] == cols[None, :], A, 0.0)), 0.5 * np.sum(w * A * Kb) / sigma]
        beta = np.atleast_1d(np.asarray(beta, float))
        if name == "rbf":
            sq = ((Xn[rows][:, None, :] - Xn[cols][None, :, :]) ** 2).sum(-1)
            out.append(0.5 * np.sum(w * A * Kb * (-0.5 * sq)))
        else:
            for k in range(d):
                sq = (Xn[rows, k][:, None] - Xn[cols, k][None, :]) ** 2
                out.append(0.5 * np.sum(w * A * Kb * (-0.5 * sq)))
        return np.array(out)

    def side_begin(self):
        pass

    def side_end(self):
        return None

    def side_join(self, token):
        pass

    def sync(self):
        pass


def _is_pd(a):
    try:
        np.linalg.cholesky(a)
        return True
    except np.linalg.LinAlgError:
        return False
