"""Host-built Gram tile tables (vgan_mmd_build_tiles, no GPU needed): every pair of the 2n x 2n matrix is
covered with the right multiplicity, every needed gradient weight is written exactly once, and the
XCD-interleaved order is a permutation."""
import numpy as np
import pytest

from vgan_amd import lib

SLOT, TWICE, STORE, MIRROR, NEG = 3, 4, 8, 16, 32


def table(n, mode, rank=0, world=1, tile=64):
    flat, cnt = lib.build_tiles(n, mode, rank, world, tile)
    return np.array(flat, dtype=np.int64).reshape(cnt, 8)


def coverage(n, tabs, nrW, wrow0s, T=64):
    """Returns (count matrix [2n,2n] of how often the sums see each pair, writes matrix).  T = 256: 256-row x 128-column tiles."""
    N = 2 * n
    TR, TC = (256, 128) if T == 256 else (T, T)
    seen = np.zeros((N, N))
    slot_ok = True
    writes = np.zeros((N, N))
    for tab, wrow0 in zip(tabs, wrow0s):
        for r0, c0, rlim, clim, fl, *_ in tab.tolist():
            ri, cj = np.arange(r0, min(r0 + TR, rlim)), np.arange(c0, min(c0 + TC, clim))
            w = 2 if fl & TWICE else 1
            seen[np.ix_(ri, cj)] += 1
            if fl & TWICE:
                seen[np.ix_(cj, ri)] += 1
            half_r, half_c = r0 >= n, c0 >= n
            want = 0 if (not half_r and not half_c) else (2 if (half_r and half_c) else 1)
            slot_ok &= (fl & SLOT) == want and bool(fl & NEG) == (half_r != half_c)
            assert rlim <= (2 * n if half_r else n) and clim <= (2 * n if half_c else n)
            if fl & STORE:
                writes[np.ix_(ri, cj)] += 1
                if fl & MIRROR:
                    writes[np.ix_(cj, ri)] += 1
            del w
    return seen, writes, slot_ok


@pytest.mark.parametrize("tile", [64, 128, 256])
@pytest.mark.parametrize("n", [1, 63, 64, 100, 128, 500, 1024])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_single_rank_tables(n, mode, tile):
    tab = table(n, mode, tile=tile)
    assert len({tuple(r[:2]) for r in tab.tolist()}) == len(tab), "duplicate tile after XCD interleave"
    seen, writes, slot_ok = coverage(n, [tab], None, [n if mode == 1 else 0], tile)
    assert slot_ok
    N = 2 * n
    want = np.ones((N, N))
    want[:n, n:] = 0  # the XY block is evaluated once, as (Y rows) x (X cols)
    assert np.array_equal(seen, want)
    if mode == 0:
        assert writes.sum() == 0
    elif mode == 1:
        assert np.array_equal(writes[n:], np.ones((n, N))) and writes[:n].sum() == 0
    else:
        assert np.array_equal(writes, np.ones((N, N)))


@pytest.mark.parametrize("tile", [64, 128, 256])
@pytest.mark.parametrize("n,world", [(128, 2), (512, 8), (1024, 4), (96, 3), (2048, 2)])
def test_row_sharded_tables(n, world, tile):
    tabs = [table(n, 1, r, world, tile) for r in range(world)]
    seen, writes, slot_ok = coverage(n, tabs, None, [0] * world, tile)
    assert slot_ok
    N = 2 * n
    want = np.ones((N, N))
    want[:n, n:] = 0
    assert np.array_equal(seen, want)          # union of the ranks = the whole matrix, no symmetry needed
    assert np.array_equal(writes[n:], np.ones((n, N))) and writes[:n].sum() == 0
    for r, tab in enumerate(tabs):             # every gradient weight a rank writes is a row of its own (mirrored ones too)
        lo, hi = n * r // world, n * (r + 1) // world
        grad = tab[(tab[:, 4] & SLOT) != 0]
        rows = grad[:, 0] - n
        assert ((rows >= lo) & (rows < hi)).all()
        mir = grad[(grad[:, 4] & MIRROR) != 0]
        assert ((mir[:, 1] - n >= lo) & (mir[:, 1] - n < hi)).all()
    # the X-X block (sums only) is shared out as its upper triangle: balanced to one tile, half the pairs of "rows x all"
    xx = [int(((t[:, 4] & SLOT) == 0).sum()) for t in tabs]
    tr, tc = (256, 128) if tile == 256 else (tile, tile)
    assert sum(xx) == sum(len(range(r, n, tc)) for r in range(0, n, tr)) and max(xx) - min(xx) <= 1


def test_bad_arguments():
    l = lib.load()
    assert l.vgan_mmd_build_tiles(0, 1, 0, 1, 64, None, 0) == -1
    assert l.vgan_mmd_build_tiles(64, 3, 0, 1, 64, None, 0) == -1
    assert l.vgan_mmd_build_tiles(64, 2, 0, 2, 64, None, 0) == -1
    assert b"grad_mode 2" in l.vgan_last_error()
    assert l.vgan_mmd_build_tiles(64, 1, 0, 1, 96, None, 0) == -1  # only 64, 128 and 256 (= 256 x 128) exist


def test_tables_property_random_shapes():
    """Property test over random (n, world, tile, mode): the union of the rank tables covers every needed pair exactly once,
    every Wg entry that the mode asks for is written exactly once, and no tile leaves its block."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(n=st.integers(1, 700), world=st.sampled_from([1, 1, 2, 3, 4, 8]), tile=st.sampled_from([64, 128, 256]),
           mode=st.sampled_from([0, 1, 2]))
    def check(n, world, tile, mode):
        if world > 1 and (mode == 2 or n < world):
            return
        if world > 1:
            mode = 1
        tabs = [table(n, mode, r, world, tile) for r in range(world)]
        seen, writes, slot_ok = coverage(n, tabs, None, [0] * world, tile)
        assert slot_ok
        N = 2 * n
        want = np.ones((N, N))
        want[:n, n:] = 0
        assert np.array_equal(seen, want)
        if mode == 0:
            assert writes.sum() == 0
        elif mode == 1:
            assert np.array_equal(writes[n:], np.ones((n, N))) and writes[:n].sum() == 0
        else:
            assert np.array_equal(writes, np.ones((N, N)))

    check()
