"""Single-process unit tests of the 2-D block-cyclic index maps (gaussian_process_amd/grid2d.py):
ownership, local offsets, slab sizes, the transposed exchange, and a NumPy simulation of one
right-looking Cholesky driven ONLY by those maps (every process's slabs and ring-free panel pieces)."""
import itertools

import numpy as np
import pytest

from gaussian_process_amd.grid2d import Grid2D, default_grid

GRIDS = [(1, 1), (2, 1), (1, 2), (2, 2), (2, 4), (4, 2), (3, 2), (2, 3)]


def test_default_grids_follow_survey_8e():
    assert [default_grid(w) for w in (1, 2, 4, 8)] == [(1, 1), (2, 1), (2, 2), (2, 4)]
    assert default_grid(6) == (2, 3) and default_grid(3) == (1, 3) and default_grid(16) == (4, 4)


@pytest.mark.parametrize("P,Q", GRIDS)
@pytest.mark.parametrize("n,nb", [(1000, 128), (1024, 128), (130, 128), (100, 128), (2049, 256)])
def test_every_lower_tile_has_exactly_one_owner_and_slot(P, Q, n, nb):
    g = Grid2D(n, nb, P, Q)
    seen = {}
    for p, q in itertools.product(range(P), range(Q)):
        rank = g.rank_of(p, q)
        assert g.coords(rank) == (p, q)
        for J in g.local_cols(q):
            rows = g.local_rows(p, J)
            assert all(I >= J and I % P == p for I in rows)
            assert g.slab_rows(p, J) == sum(g.tile_h(I) for I in rows)
            for I in rows:
                assert g.owner(I, J) == rank
                assert (I, J) not in seen
                seen[(I, J)] = rank
    assert set(seen) == {(I, J) for I in range(g.T) for J in range(I + 1)}
    # local offsets are consistent with the local lists
    for p in range(P):
        rows = g.local_rows(p)
        assert [g.lrow(I) for I in rows] == [i * nb for i in range(len(rows))]
        assert g.nlr(p) == len(rows)
    for q in range(Q):
        cols = g.local_cols(q)
        assert [g.lcol(J) for J in cols] == [i * nb for i in range(len(cols))]
    # diagonal owners
    for k in range(g.T):
        p, q = k % P, k % Q
        assert k in g.diag_tiles(p, q)
    assert sum(len(g.diag_tiles(p, q)) for p in range(P) for q in range(Q)) == g.T


@pytest.mark.parametrize("P,Q", GRIDS)
def test_resident_matrix_is_the_lower_triangle_over_the_grid(P, Q):
    """Per-process storage ~ lower triangle / (P Q): the excess is the upper halves of the diagonal
    tiles and the block-cyclic imbalance, both O(n nb)."""
    n, nb = 65536, 512
    g = Grid2D(n, nb, P, Q)
    ideal = g.lower_elems() / (P * Q)
    for p, q in itertools.product(range(P), range(Q)):
        e = g.slab_elems(p, q)
        assert e <= ideal * 1.05 + 2 * n * nb, (p, q, e / ideal)
        assert e >= ideal * 0.93, (p, q, e / ideal)
    total = sum(g.slab_elems(p, q) for p in range(P) for q in range(Q))
    assert total == sum(g.tile_h(I) * nb for I in range(g.T) for J in range(I + 1))


@pytest.mark.parametrize("P,Q", GRIDS)
def test_transposed_pick_lists_exactly_the_column_tiles(P, Q):
    g = Grid2D(3000, 128, P, Q)
    for k in range(g.T - 1):
        for q in range(Q):
            want = g.local_cols(q, k + 1)
            got = []
            for p in range(P):
                rows = g.local_rows(p, k + 1)
                t0, step, cnt = g.transposed_pick(p, q, k)
                assert cnt <= g.transposed_max(q, k)
                got += [rows[t0 + i * step] for i in range(cnt)]
            assert sorted(got) == want


@pytest.mark.parametrize("P,Q", [(1, 1), (2, 1), (2, 2), (2, 4), (3, 2)])
def test_simulated_cholesky_through_the_maps(P, Q):
    """Right-looking tile Cholesky in which every process touches only the tiles grid2d gives it and
    the panel pieces the row broadcast / transposed exchange would deliver."""
    rng = np.random.default_rng(0)
    n, nb = 700, 128
    g = Grid2D(n, nb, P, Q)
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    ref = np.linalg.cholesky(A)
    tile = lambda M, I, J: M[I * nb: I * nb + g.tile_h(I), J * nb: J * nb + g.tile_h(J)]
    local = {(p, q): {(I, J): tile(A, I, J).copy() for J in g.local_cols(q) for I in g.local_rows(p, J)}
             for p in range(P) for q in range(Q)}
    for k in range(g.T):
        pk, qk = k % P, k % Q
        Lkk = np.linalg.cholesky(np.tril(local[(pk, qk)][(k, k)]) + np.tril(local[(pk, qk)][(k, k)], -1).T)
        local[(pk, qk)][(k, k)] = Lkk
        panel = {}
        for p in range(P):                                   # column owners solve their rows
            for I in g.local_rows(p, k + 1):
                local[(p, qk)][(I, k)] = np.linalg.solve(Lkk, local[(p, qk)][(I, k)].T).T
                panel[I] = local[(p, qk)][(I, k)]
        for p in range(P):
            for q in range(Q):
                rowset = {I: panel[I] for I in g.local_rows(p, k + 1)}                     # row broadcast
                colset = {}
                for src in range(P):                                                          # transposed exchange
                    rows = g.local_rows(src, k + 1)
                    t0, step, cnt = g.transposed_pick(src, q, k)
                    for i in range(cnt):
                        colset[rows[t0 + i * step]] = panel[rows[t0 + i * step]]
                for J in g.local_cols(q, k + 1):
                    for I in g.local_rows(p, J):
                        local[(p, q)][(I, J)] -= rowset[I] @ colset[J].T
    out = np.zeros((n, n))
    for tiles in local.values():
        for (I, J), v in tiles.items():
            tile(out, I, J)[...] = np.tril(v) if I == J else v
    assert np.abs(out - ref).max() < 1e-10 * np.abs(ref).max()
