"""Replay model of the fused formation + factorization launch (form_factor.h): given the ORDERED work list, simulate the
ticket draw (in order, a worker draws when it is free), the blocking waits of every item and the pivot chain beside them,
with durations calibrated on an item trace (tools/ff_trace.py, profiles/r04_ff_trace_*.txt).  CPU only: the tool the list
generator (ff_schedule.h) is tuned with.  `python tools/ff_replay.py [trace.npz]` compares the model with a trace."""
import heapq
import sys

import numpy as np

FF_F, FF_T = 0, 1
INIT, ADD_BASE, PANEL, SIG0 = 1, 2, 4, 8


class Model:
    """chain_mode 0 (three launches per chain step beside 224 workers), calibrated on profiles/r04_ff_item_trace_baseline.txt"""
    f_over, f_stage = 17.9, 3.91          # F chunk: us fixed + per BK=16 stage
    t_over, t_col, t_base, t_panel, t_rmw = 4.1, 15.8, 53.0, 21.0, 2.0     # t_base: FOUR slabs
    gap = 0.8                             # end of an item -> next ticket drawn
    d_item = 140.0
    potrf, cpanel, cupdate = 36.0, 8.0, 5.0
    g_potrf_panel, g_panel_update, g_update_potrf = 3.5, 3.5, 4.0
    chain_start = 220.0                   # ff_maxdiag in front of potrf(0)
    handoff = 1.5                         # counter visible to a spinning consumer


class ModelRoles(Model):
    """chain_mode 1 (the chain as roles of the one launch, 251 workers), calibrated on profiles/r04_ff_item_trace_roles_kernel.txt"""
    f_over, f_stage = 6.1, 4.02
    t_over, t_col, t_base, t_panel, t_rmw = 7.0, 15.9, 27.0, 18.3, 1.0
    d_item = 397.0
    potrf, cpanel, cupdate = 36.8, 7.3, 7.3
    g_potrf_panel, g_panel_update, g_update_potrf = 1.0, 1.5, 1.0
    chain_start = 0.0
    handoff = 1.0


def tile_id(i, c):
    return i * (i + 1) // 2 + c


def replay(items, nblk, W=224, M=Model, tile_q=None, verbose=False, mode=0):
    """items: (n, 8) uint8 FFItem records.  Returns dict(end, chain potrf start times, worker busy/wait sums, per-item times)."""
    n = len(items)
    typ = items[:, 0]; ti = items[:, 1].astype(int); tc = items[:, 2].astype(int)
    j0 = items[:, 4].astype(int); j1 = items[:, 5].astype(int); fl = items[:, 6].astype(int); seq = items[:, 7].astype(int)
    s0 = items[:, 4].astype(int) + 256 * items[:, 5].astype(int); s1 = items[:, 6].astype(int) + 256 * items[:, 7].astype(int)
    ntile = nblk * (nblk + 1) // 2
    INF = float("inf")
    # tile_q: chunks per tile (from the list itself)
    fq = np.zeros(ntile, int)
    for k in np.nonzero(typ == FF_F)[0]:
        if tc[k] <= ti[k]:
            fq[tile_id(ti[k], tc[k])] += 1
        if ti[k] + 1 < nblk:
            fq[tile_id(ti[k] + 1, tc[k])] += 1
    titems = np.zeros(ntile, int)
    for k in np.nonzero(typ == FF_T)[0]:
        titems[tile_id(ti[k], tc[k])] += 1
    # event-driven: each resource is resolved lazily through "time at which X becomes true" tables filled as items finish.
    fdone_cnt = np.zeros(ntile, int); fdone_time = np.zeros(ntile)           # time the LAST chunk so far finished
    form_time = np.full(ntile, INF)
    tprog_time = {}                                                          # (tile, seq) -> time
    lfin = [np.full(r + 1, INF) for r in range(nblk)]                        # lfin[r][c]: time tile (r,c) became L (c < r); [r][r] unused
    potrf_done = np.full(nblk, INF); potrf_start = np.full(nblk, INF)
    dready = np.full(nblk, INF)
    t_draw = np.zeros(n); t_ready = np.zeros(n); t_end = np.zeros(n); who = np.zeros(n, int)
    free = [(0.0, w) for w in range(W)]
    heapq.heapify(free)
    # The chain depends on items and items on the chain: process items in list order (each item's start only depends on
    # EARLIER items and on the chain, which only depends on earlier items) and advance the chain lazily.
    chain = {"k": 0, "phase": 0, "free": M.chain_start if mode == 0 else 0.0}
    dcount = {"n": 0, "t": 0.0}

    def advance_chain():
        while chain["k"] < nblk:
            k = chain["k"]
            if chain["phase"] == 0:
                if dready[k] == INF:
                    return
                if mode and k == 0 and dcount["n"] < nblk:
                    return
                st = max(chain["free"], dready[k] + M.handoff, dcount["t"] + M.handoff if mode else 0.0)
                potrf_start[k] = st
                potrf_done[k] = st + M.potrf
                chain["free"] = potrf_done[k] + M.g_potrf_panel
                if k + 1 >= nblk:
                    chain["k"] = nblk
                    return
                chain["phase"] = 1
            if chain["phase"] == 1:
                t = tprog_time.get((tile_id(k + 1, k), titems[tile_id(k + 1, k)]), INF)
                if titems[tile_id(k + 1, k)] == 0:
                    t = form_time[tile_id(k + 1, k)]
                if t == INF:
                    return
                e = max(chain["free"], t + M.handoff) + M.cpanel
                lfin[k + 1][k] = e
                chain["free"] = e + M.g_panel_update
                chain["phase"] = 2
            if chain["phase"] == 2:
                t = tprog_time.get((tile_id(k + 1, k + 1), titems[tile_id(k + 1, k + 1)]), INF)
                if t == INF:
                    return
                e = max(chain["free"], t + M.handoff) + M.cupdate
                dready[k + 1] = e
                chain["free"] = e + M.g_update_potrf
                chain["k"] = k + 1
                chain["phase"] = 0

    def lfinal_time(r, upto):
        """time at which the leading `upto` tiles of row r are final L"""
        if upto <= 0:
            return 0.0
        return max(lfin[r][:upto])

    for k in range(n):
        t0, w = heapq.heappop(free)
        t_draw[k] = t0; who[k] = w
        if typ[k] == 2:                      # FF_D
            e = t0 + M.d_item
            dcount["n"] += 1; dcount["t"] = max(dcount["t"], e)
            t_ready[k] = t0; t_end[k] = e
            heapq.heappush(free, (e + M.gap, w))
            continue
        if typ[k] == FF_F:
            e = t0 + M.f_over + M.f_stage * max(0, s1[k] - s0[k])
            for r in (ti[k], ti[k] + 1):
                if r >= tc[k] and r < nblk:
                    t = tile_id(r, tc[k])
                    fdone_cnt[t] += 1; fdone_time[t] = max(fdone_time[t], e)
                    if fdone_cnt[t] == fq[t]:
                        form_time[t] = fdone_time[t]
            t_ready[k] = t0; t_end[k] = e
            heapq.heappush(free, (e + M.gap, w))
            continue
        t = tile_id(ti[k], tc[k])
        rdy = t0
        if fl[k] & ADD_BASE:
            rdy = max(rdy, form_time[t] + M.handoff)
        if not (fl[k] & INIT):
            rdy = max(rdy, tprog_time.get((t, seq[k] - 1), INF) + M.handoff)
        if j1[k] > j0[k]:
            advance_chain()
            rdy = max(rdy, lfinal_time(ti[k], j1[k]) + M.handoff)
            if tc[k] != ti[k]:
                rdy = max(rdy, lfinal_time(tc[k], j1[k]) + M.handoff)
        if rdy == INF:
            raise RuntimeError("item %d (T %d,%d [%d,%d) flags %d) waits for something later in the list" % (k, ti[k], tc[k], j0[k], j1[k], fl[k]))
        e = rdy + M.t_over + M.t_col * (j1[k] - j0[k]) + (M.t_base * fq[t] / 4.0 if fl[k] & ADD_BASE else 0.0) + (0.0 if fl[k] & INIT else M.t_rmw)
        if fl[k] & PANEL:
            advance_chain()
            if potrf_done[tc[k]] == INF:
                raise RuntimeError("panel item %d waits for potrf(%d) which the chain cannot reach" % (k, tc[k]))
            e = max(e, potrf_done[tc[k]] + M.handoff) + M.t_panel
            lfin[ti[k]][tc[k]] = e
        tprog_time[(t, seq[k])] = e
        if fl[k] & SIG0:
            dready[ti[k]] = e
        t_ready[k] = rdy; t_end[k] = e
        heapq.heappush(free, (e + M.gap, w))
    advance_chain()
    T = typ == FF_T
    Fm = typ == FF_F
    return {"end": max(potrf_done[nblk - 1], t_end.max()), "chain_end": potrf_done[nblk - 1], "potrf_start": potrf_start,
            "form_end": t_end[Fm].max(), "wait_sum": float((t_ready[T] - t_draw[T]).sum()),
            "t_work": float((t_end[T] - t_ready[T]).sum()), "f_work": float((t_end[Fm] - t_draw[Fm]).sum()),
            "t_draw": t_draw, "t_ready": t_ready, "t_end": t_end, "who": who}


def summary(r, W=224):
    return "end %.0f us (chain %.0f, formation %.0f); per worker: F %.0f, T %.0f, wait %.0f us" % (
        r["end"], r["chain_end"], r["form_end"], r["f_work"] / W, r["t_work"] / W, r["wait_sum"] / W)


if __name__ == "__main__":
    f = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_ff_trace0.npz"
    z = np.load(f)
    items, nblk, nit = z["items"], int(z["nblk"]), int(z["nit"])
    r = replay(items, nblk)
    print("model:", summary(r))
    tr = z["trace"]
    it = tr[:4 * nit].reshape(nit, 4).astype(float)
    ch = tr[4 * nit:].reshape(-1, 12).astype(float)
    t0 = it[:, 0][it[:, 0] > 0].min()
    print("trace: chain end %.0f, formation end %.0f" % ((ch[nblk - 1, 2] - t0) / 100, (it[items[:, 0] == 0, 2].max() - t0) / 100))
    print("potrf start, model vs trace:")
    for k in range(nblk):
        print("  %2d  %7.0f  %7.0f" % (k, r["potrf_start"][k], (ch[k, 1] - t0) / 100))
