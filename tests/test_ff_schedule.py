"""CPU tests of the work list of the fused formation + factorization (csrc/ff_schedule.h, exported host-only through
ipm_debug_ff_schedule): the list is replayed IN ORDER against the dependency rules the device kernel waits on
(csrc/form_factor.h) with the pivot chain advanced as far as its inputs allow before every item.  An item whose inputs
are not all produced by EARLIER items (or by chain steps that themselves only need earlier items) could leave a worker
waiting for a ticket nobody holds -- the one way the persistent launch could hang -- so that is what is asserted, together
with completeness: every tile receives its formation chunks, its base exactly once, every column of L up to its limit
exactly once and in order, and its panel solve exactly once, last."""
import ctypes as C

import numpy as np
import pytest

from interiorpointmethod_amd import _lib

F, T, D = 0, 1, 2
NSTAGES = 512          # BK = 16 stages of the K = 8192 formation the debug entry point models
INIT, ADD_BASE, PANEL, SIG_DIAG0 = 1, 2, 4, 8


def schedule(nblk, q, workers):
    lib = _lib.load()
    cap = 200000
    items = (C.c_ubyte * (8 * cap))()
    count = C.c_int32(0)
    ntile = nblk * (nblk + 1) // 2
    tile_items = (C.c_int32 * ntile)()
    sim = (C.c_double * 2)()
    rc = lib.ipm_debug_ff_schedule(nblk, q, workers, items, cap, C.byref(count), tile_items, sim)
    assert rc == 0 and 0 < count.value <= cap
    raw = np.frombuffer(items, dtype=np.uint8)[:8 * count.value].reshape(-1, 8)
    arr = raw.astype(int)
    # F items carry their stage range [s0, s1) as two uint16 in the last four bytes: columns 8, 9 of the table
    st = raw[:, 4:8].copy().view(np.uint16).astype(int)
    arr = np.concatenate([arr, st], axis=1)
    return arr, np.array(tile_items[:]), (sim[0], sim[1])


def tid(i, c):
    return i * (i + 1) // 2 + c


def replay(nblk, q, items, tile_items):
    """Replay the list in order; returns the number of chain steps completed at the end."""
    ntile = nblk * (nblk + 1) // 2
    fcount = np.zeros(ntile, int)
    base = np.zeros(ntile, int)
    applied = np.zeros(ntile, int)
    paneled = np.zeros(ntile, int)
    nit = np.zeros(ntile, int)
    final = [[False] * (r + 1) for r in range(nblk)]            # final[r][j]: tile (r, j) is L
    potrf = [False] * nblk
    chain = {"k": 0, "phase": 0}
    limit = lambda i, c: (max(c - 1, 0) if i == c else c)                          # noqa: E731
    needs_panel = lambda i, c: i > c + 1                                           # noqa: E731
    done = lambda i, c: base[tid(i, c)] == 1 and applied[tid(i, c)] == limit(i, c) and nit[tid(i, c)] == tile_items[tid(i, c)]   # noqa: E731

    def advance():
        while chain["k"] < nblk:
            k = chain["k"]
            if chain["phase"] == 0:
                if k == 0 and not done(0, 0):
                    return
                potrf[k] = True                                    # (k > 0: its diagonal tile was completed by phase 2 of step k-1)
                if k + 1 >= nblk:
                    chain["k"] = nblk
                    return
                chain["phase"] = 1
            if chain["phase"] == 1:
                if not done(k + 1, k):
                    return
                final[k + 1][k] = True
                chain["phase"] = 2
            if chain["phase"] == 2:
                if not done(k + 1, k + 1):
                    return
                chain["k"], chain["phase"] = k + 1, 0

    fcover = {}
    # chunks per tile: q for ordinary tiles, more (shorter ones) for the block rows the chain needs first
    qtile = np.zeros(ntile, int)
    for (typ, i, c, qq, *_r) in items:
        if typ == F:
            for ii in (i, i + 1):
                if c <= ii < nblk:
                    qtile[tid(ii, c)] = max(qtile[tid(ii, c)], qq + 1)
    assert qtile.min() >= 1 and np.all(qtile[tid(nblk - 1, 0):] == min(q, 512)) or nblk <= 8
    for n, (typ, i, c, qq, j0, j1, flags, seq, s0, s1) in enumerate(items):
        if typ == D:                         # max diag(B) of row block i: the items the chain-in-kernel launch starts with
            assert n < nblk and i == n
            continue
        advance()
        if typ == T:
            assert c <= i < nblk
        t = tid(i, c) if c <= i else -1
        if typ == F:
            # a formation item covers the tile PAIR (i, c), (i + 1, c), i even (a half above the diagonal / below the matrix is dropped)
            assert i % 2 == 0 and c <= min(i + 1, nblk - 1) and 0 <= qq < 16 and 0 <= s0 <= s1 <= NSTAGES
            for ii in (i, i + 1):
                if c <= ii < nblk:
                    tt = tid(ii, c)
                    assert (tt, qq) not in fcover
                    fcover[(tt, qq)] = (s0, s1)
                    fcount[tt] += 1
            continue
        assert typ == T
        assert seq == nit[t] + 1, (n, i, c, "sequence")
        assert bool(flags & INIT) == (nit[t] == 0), (n, i, c, "INIT on the first item only")
        assert j0 == applied[t] and j0 <= j1 <= limit(i, c), (n, i, c, j0, j1)
        if flags & ADD_BASE:
            assert fcount[t] == qtile[t] and base[t] == 0, (n, i, c, "base before its chunks, or twice")
            base[t] = 1
        for j in range(j0, j1):
            assert final[i][j] and final[c][j], (n, i, c, j, "operand tile not final at this point of the list")
        applied[t] = j1
        if flags & PANEL:
            assert needs_panel(i, c) and base[t] == 1 and applied[t] == limit(i, c) and not paneled[t], (n, i, c)
            assert potrf[c], (n, i, c, "panel solve before its diagonal block is factored")
            paneled[t] = 1
            final[i][c] = True
        assert bool(flags & SIG_DIAG0) == (i == 0 and c == 0)
        nit[t] += 1
        assert not (paneled[t] and not (flags & PANEL)), (n, i, c, "an item after the panel solve")
    advance()
    for i in range(nblk):
        for c in range(i + 1):
            t = tid(i, c)
            cuts = sorted(fcover[(t, k)] for k in range(qtile[t]))    # the chunks tile the K loop exactly once
            assert cuts[0][0] == 0 and cuts[-1][1] == NSTAGES and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])), (i, c, cuts)
            assert fcount[t] == qtile[t] and base[t] == 1 and applied[t] == limit(i, c), (i, c)
            assert paneled[t] == (1 if needs_panel(i, c) else 0) and nit[t] == tile_items[t] >= 1, (i, c)
    return chain["k"]


@pytest.mark.parametrize("nblk,q,workers", [(3, 2, 4), (4, 4, 248), (8, 4, 31), (16, 4, 248), (32, 4, 248), (32, 8, 248),
                                            (32, 4, 120), (33, 3, 248), (64, 4, 248)])
def test_work_list_is_complete_and_every_item_follows_what_it_needs(built_lib, nblk, q, workers):
    items, tile_items, sim = schedule(nblk, q, workers)
    assert replay(nblk, q, items, tile_items) == nblk
    assert sim[0] > 0 and sim[1] > 0


def test_work_list_is_deterministic_and_its_simulated_time_beats_the_serial_path(built_lib):
    a, ta, sa = schedule(32, 4, 248)
    b, tb, sb = schedule(32, 4, 248)
    assert np.array_equal(a, b) and np.array_equal(ta, tb) and sa == sb
    # the model's own estimate at the headline size: formation + factorization well under the 4.3 ms of the serial path
    assert sa[0] < 4100.0, sa
    n_t = int((a[:, 0] == T).sum())
    assert n_t < 6000          # batching keeps the read-modify-write passes per tile small (pure right-looking: 5456 + 528)
