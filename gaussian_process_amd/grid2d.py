"""Index maps of the 2-D block-cyclic layout of the multi-GPU path.

The n x n lower-triangular matrix (Ky, then its Cholesky factor, regression.py:36,49) is cut into
nb x nb tiles; tile (I, J), I >= J, lives on the process with grid coordinates (I mod P, J mod Q) of a
P x Q process grid, rank = p * Q + q.  A process keeps, for every tile column J it owns (J mod Q == q),
ONE contiguous slab holding its tiles of that column from the diagonal down (rows I >= J, I mod P == p),
nb doubles wide: the strictly upper tiles are never stored, so the resident matrix is the lower
triangle divided by P * Q plus the diagonal tiles' upper halves.

Pure host arithmetic, no torch / GPU: unit-tested single-process (tests/test_grid2d.py).
"""
from math import gcd


def default_grid(world):
    """P x Q for `world` processes: P the largest divisor of `world` not above its square root, Q the
    rest -- 1x1, 2x2, 2x4 -- except two processes, which form a 2x1 grid so that the panel
    solve of every step is shared by both (a 1x2 grid would leave it to one of them)."""
    if world == 2:
        return 2, 1
    p = 1
    for c in range(1, int(world ** 0.5) + 1):
        if world % c == 0:
            p = c
    return p, world // p


class Grid2D:
    def __init__(self, n, nb, P, Q):
        assert n >= 1 and nb >= 1 and P >= 1 and Q >= 1
        self.n, self.nb, self.P, self.Q = n, nb, P, Q
        self.T = (n + nb - 1) // nb                       # tiles per dimension
        self.L = P * Q // gcd(P, Q)                       # period of the diagonal owners

    # ---- processes ------------------------------------------------------------------------
    def coords(self, rank):
        return divmod(rank, self.Q)

    def rank_of(self, p, q):
        return p * self.Q + q

    def row_ranks(self, p):
        """Ranks of process row p (they share the tile rows I mod P == p)."""
        return [self.rank_of(p, q) for q in range(self.Q)]

    def col_ranks(self, q):
        """Ranks of process column q (they share the tile columns J mod Q == q)."""
        return [self.rank_of(p, q) for p in range(self.P)]

    def owner(self, I, J):
        return self.rank_of(I % self.P, J % self.Q)

    # ---- tiles ----------------------------------------------------------------------------
    def tile_h(self, I):
        """Rows (= columns) of tile row (column) I: nb, except the ragged last one."""
        return min(self.nb, self.n - I * self.nb)

    def first_at_or_after(self, t, r, m):
        """Smallest index >= t congruent to r modulo m."""
        return t + ((r - t) % m)

    def local_rows(self, p, start=0):
        """Tile rows I >= start owned by process row p, ascending."""
        return list(range(self.first_at_or_after(start, p, self.P), self.T, self.P))

    def local_cols(self, q, start=0):
        return list(range(self.first_at_or_after(start, q, self.Q), self.T, self.Q))

    def lrow(self, I):
        """Element offset of tile row I in the local row space of its process row (rows of the
        workspaces indexed by local row: I // P tiles of nb rows come before it)."""
        return (I // self.P) * self.nb

    def lcol(self, J):
        return (J // self.Q) * self.nb

    def nlr(self, p):
        """Number of tile rows process row p owns."""
        return len(range(p, self.T, self.P))

    def nlc(self, q):
        return len(range(q, self.T, self.Q))

    def rows_from(self, p, start):
        """Matrix rows held by process row p in tile rows >= start."""
        return sum(self.tile_h(I) for I in self.local_rows(p, start))

    def cols_from(self, q, start):
        return sum(self.tile_h(J) for J in self.local_cols(q, start))

    # ---- slabs (the resident matrix) ---------------------------------------------------------
    def slab_rows(self, p, J):
        """Rows of the slab process (p, J mod Q) keeps for tile column J: its tile rows I >= J."""
        return self.rows_from(p, J)

    def slab_elems(self, p, q):
        """Doubles of matrix storage on process (p, q): every slab is nb wide."""
        return sum(self.slab_rows(p, J) * self.nb for J in self.local_cols(q))

    def lower_elems(self):
        """Doubles of the lower triangle including the diagonal."""
        return self.n * (self.n + 1) // 2

    def diag_tiles(self, p, q, start=0):
        """Diagonal tiles (k, k), k >= start, owned by process (p, q): k = p (mod P) and k = q (mod Q)."""
        return [k for k in range(start, self.T) if k % self.P == p and k % self.Q == q]

    # ---- the transposed exchange of a panel -----------------------------------------------------
    def transposed_pick(self, p, q, k):
        """Of the tiles I > k of panel k that process row p holds after the row broadcast (I mod P == p,
        ascending, position t = 0, 1, ...), the positions of those process column q needs as COLUMN
        tiles (I mod Q == q): an arithmetic progression (t0, step, count), count = 0 if none."""
        rows = self.local_rows(p, k + 1)
        pos = [t for t, I in enumerate(rows) if I % self.Q == q]
        if not pos:
            return 0, 1, 0
        step = self.L // self.P
        assert all(b - a == step for a, b in zip(pos[:-1], pos[1:]))
        return pos[0], step, len(pos)

    def transposed_max(self, q, k):
        """Largest per-source tile count of the exchange in process column q (padding of the all-gather)."""
        return max(self.transposed_pick(p, q, k)[2] for p in range(self.P))
