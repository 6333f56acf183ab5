"""Prototype of the work-list generator of the fused formation + factorization (candidate for ff_schedule.h): list scheduling of
the item DAG by bottom level (longest path to the end of the factorization) on the calibrated durations of tools/ff_replay.py.
CPU only.  python tools/ff_gen.py [batch] [Q]"""
import heapq
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tools.ff_replay import Model, replay, summary, tile_id, FF_F, FF_T, INIT, ADD_BASE, PANEL, SIG0   # noqa: E402


def limit(i, c):
    return (c - 1 if c > 0 else 0) if i == c else c


def needs_panel(i, c):
    return i > c + 1


def build(nblk=32, W=224, Q=4, batch=4, tail=2, stagger=0.0, nstages=512, M=Model, base_on="pre_final", q_of=None, prio_mode="bl", look=10.0, verbose=False):
    """-> items (n, 8) uint8 in ticket order"""
    INF = float("inf")
    nodes = []          # dict(kind, dur, succ[], npred, ...)

    def add(**kw):
        kw.setdefault("succ", []); kw["npred"] = 0; kw["id"] = len(nodes)
        nodes.append(kw)
        return kw["id"]

    def edge(a, b):
        nodes[a]["succ"].append(b); nodes[b]["npred"] += 1

    # ---- chain
    potrf = [add(kind="potrf", k=k, dur=M.potrf + M.g_potrf_panel) for k in range(nblk)]
    cpan = [add(kind="cpanel", k=k, dur=M.cpanel + M.g_panel_update) for k in range(nblk - 1)]
    cupd = [add(kind="cupdate", k=k, dur=M.cupdate + M.g_update_potrf) for k in range(nblk - 1)]
    for k in range(nblk - 1):
        edge(potrf[k], cpan[k]); edge(cpan[k], cupd[k]); edge(cupd[k], potrf[k + 1])
    # ---- T items per tile
    titems = {}         # (i,c) -> list of node ids in sequence
    lprod = {}          # (r,c) -> node that makes L(r,c) final (c < r)
    for i in range(nblk):
        for c in range(i + 1):
            lim = limit(i, c)
            cuts = []
            pan = needs_panel(i, c)
            # bulk batches over [0, lim - tail), then `tail` single columns: the newest columns are applied one at a time
            # (a batch that waits for column j cannot start before L(., j) exists, one pipeline step before the final item is due)
            nb_end = max(0, lim - tail)
            j = 0
            while j + batch <= nb_end:
                cuts.append((j, j + batch)); j += batch
            if j < nb_end:
                cuts.append((j, nb_end)); j = nb_end
            while j < lim:
                cuts.append((j, j + 1)); j += 1
            if not cuts:
                cuts.append((0, 0))
            ids = []
            for s, (a, b) in enumerate(cuts):
                last = s == len(cuts) - 1
                ids.append(add(kind="T", i=i, c=c, j0=a, j1=b, seq=s + 1, flags=(INIT if s == 0 else 0) | (PANEL if (last and pan) else 0) | (SIG0 if (i == 0 and c == 0) else 0),
                               dur=M.t_over + M.t_col * (b - a) + (0 if s == 0 else M.t_rmw) + (M.t_panel if (last and pan) else 0.0) + M.handoff + M.gap))
                if s:
                    edge(ids[-2], ids[-1])
            titems[(i, c)] = ids
            if pan:
                lprod[(i, c)] = ids[-1]
    for k in range(nblk - 1):
        lprod[(k + 1, k)] = cpan[k]
        edge(titems[(k + 1, k)][-1], cpan[k])
        edge(titems[(k + 1, k + 1)][-1], cupd[k])
    edge(titems[(0, 0)][-1], potrf[0])
    for (i, c), ids in titems.items():
        for nid in ids:
            nd = nodes[nid]
            if nd["j1"] > nd["j0"]:
                edge(lprod[(i, nd["j1"] - 1)], nid)
                if c != i:
                    edge(lprod[(c, nd["j1"] - 1)], nid)
            if nd["flags"] & PANEL:
                edge(potrf[c], nid)
    # ---- base: which T item of the tile adds the formation slabs
    base_item = {}
    for (i, c), ids in titems.items():
        if base_on == "first" or len(ids) == 1:
            b = ids[0]
        elif base_on == "pre_final":
            b = ids[-2]
        elif base_on == "last_bulk":
            k_ = len(ids) - 1
            while k_ > 0 and nodes[ids[k_]]["j1"] - nodes[ids[k_]]["j0"] <= 1:
                k_ -= 1
            b = ids[k_]
        else:
            b = ids[-1]
        base_item[(i, c)] = b
    # ---- F chunks (pairs)
    fch = {}
    qof = q_of or (lambda i, c: Q)
    npairs = 0
    for i in range(0, nblk, 2):
        for c in range(min(i + 1, nblk - 1) + 1):
            q = qof(i, c)
            # chunk lengths spread over (1 - stagger) .. (1 + stagger) of the mean, phase per pair: the workers do not finish their
            # formation chunks in lockstep (with equal chunks all of them are deaf for a whole chunk at the same time)
            phi = (npairs * 0.381966) % 1.0
            npairs += 1
            lens = [1.0 + stagger * (2.0 * (((k / q) + phi) % 1.0) - 1.0) for k in range(q)]
            tot = sum(lens); acc = 0.0; cutp = [0]
            for k in range(q):
                acc += lens[k]; cutp.append(int(nstages * acc / tot + 0.5))
            cutp[-1] = nstages
            for k in range(q):
                s0, s1 = cutp[k], cutp[k + 1]
                nid = add(kind="F", i=i, c=c, q=k, s0=s0, s1=s1, dur=M.f_over + M.f_stage * (s1 - s0) + M.gap)
                fch[(i, c, k)] = nid
                for r in (i, i + 1):
                    if r >= c and r < nblk:
                        edge(nid, base_item[(r, c)])
    tq = {}
    for (i, c, k) in fch:
        for r in (i, i + 1):
            if r >= c and r < nblk:
                tq[(r, c)] = tq.get((r, c), 0) + 1
    for (i, c), b in base_item.items():
        nodes[b]["flags"] |= ADD_BASE
        nodes[b]["dur"] += M.t_base * tq[(i, c)] / 4.0
    # ---- bottom levels (reverse topological order)
    order, indeg = [], [nd["npred"] for nd in nodes]
    stack = [nd["id"] for nd in nodes if nd["npred"] == 0]
    while stack:
        u = stack.pop(); order.append(u)
        for v in nodes[u]["succ"]:
            indeg[v] -= 1
            if indeg[v] == 0:
                stack.append(v)
    assert len(order) == len(nodes), "cycle"
    bl = [0.0] * len(nodes)
    for u in reversed(order):
        bl[u] = nodes[u]["dur"] + max([bl[v] for v in nodes[u]["succ"]], default=0.0)
    # ---- list scheduling: workers draw in order; a worker that frees at t takes the highest-priority item among those whose
    #      predecessors are all SCHEDULED and whose inputs are ready by t + look (it waits for them inside), else a formation chunk
    fin = [INF] * len(nodes)                 # finish time once scheduled
    est = [0.0] * len(nodes)                 # max finish of scheduled predecessors
    left = [nd["npred"] for nd in nodes]
    avail_T = []                             # heap of (-prio, id) of T items with all predecessors scheduled
    avail_F = [(-bl[n], n) for n in fch.values()]
    heapq.heapify(avail_F)
    chain_ready = [n for n in potrf + cpan + cupd if left[n] == 0]

    def release(u):
        for v in nodes[u]["succ"]:
            est[v] = max(est[v], fin[u])
            left[v] -= 1
            if left[v] == 0:
                if nodes[v]["kind"] == "T":
                    heapq.heappush(avail_T, (-bl[v], v))
                elif nodes[v]["kind"] != "F":
                    # chain nodes run by themselves as soon as their inputs are there
                    st = est[v] if nodes[v]["kind"] != "potrf" or nodes[v]["k"] > 0 else max(est[v], M.chain_start)
                    fin[v] = st + nodes[v]["dur"]
                    release(v)

    for n in list(titems[(0, 0)][:1]):
        pass
    for (i, c), ids in titems.items():
        if left[ids[0]] == 0:
            heapq.heappush(avail_T, (-bl[ids[0]], ids[0]))
    free = [(0.0, w) for w in range(W)]
    heapq.heapify(free)
    out = []
    nT = sum(len(v) for v in titems.values())
    done_T = 0
    while done_T < nT or avail_F:
        t, w = heapq.heappop(free)
        # best T item that is ready soon enough
        pick = None
        skipped = []
        while avail_T:
            p, v = heapq.heappop(avail_T)
            if est[v] <= t + look:
                pick = v
                break
            skipped.append((p, v))
        best_skipped = skipped[0] if skipped else None
        for x in skipped:
            heapq.heappush(avail_T, x)
        if pick is not None and avail_F and -avail_F[0][0] > bl[pick] and prio_mode == "bl":
            # a formation chunk is more urgent than the best ready update
            heapq.heappush(avail_T, (-bl[pick], pick)); pick = None
        if pick is None and avail_F:
            _, v = heapq.heappop(avail_F)
            fin[v] = t + nodes[v]["dur"]
            out.append(v)
            release(v)
            heapq.heappush(free, (fin[v], w))
            continue
        if pick is None:
            if not avail_T:
                # nothing schedulable yet: this worker sleeps until the next item becomes available (a chain node may release it)
                heapq.heappush(free, (t + 5.0, w))
                continue
            # formation exhausted: take the earliest-ready item and wait inside it
            v = min(avail_T, key=lambda x: (est[x[1]], x[0]))
            avail_T.remove(v); heapq.heapify(avail_T)
            pick = v[1]
        st = max(t, est[pick])
        fin[pick] = st + nodes[pick]["dur"]
        out.append(pick)
        done_T += 1
        release(pick)
        heapq.heappush(free, (fin[pick], w))
    # ---- encode
    items = np.zeros((len(out), 8), dtype=np.uint8)
    for n, v in enumerate(out):
        nd = nodes[v]
        if nd["kind"] == "F":
            items[n] = [FF_F, nd["i"], nd["c"], nd["q"], nd["s0"] & 255, nd["s0"] >> 8, nd["s1"] & 255, nd["s1"] >> 8]
        else:
            items[n] = [FF_T, nd["i"], nd["c"], 0, nd["j0"], nd["j1"], nd["flags"], nd["seq"]]
    gen_end = max(f for f in fin if f != INF)
    return items, gen_end


if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for base_on in ("last_bulk", "pre_final"):
        for stagger in (0.0, 0.3, 0.6):
            for look in (0.0, 40.0):
                items, gen_end = build(batch=batch, Q=Q, base_on=base_on, tail=2, look=look, stagger=stagger)
                r = replay(items, 32)
                print("batch %d Q %d base %-9s stagger %.1f look %2.0f | items %d | generator %.0f | replay: %s" % (batch, Q, base_on, stagger, look, len(items), gen_end, summary(r)))
