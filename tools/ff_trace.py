"""Time line of ONE fused formation + factorization launch (IPM_FF_TRACE_ITEMS=1): where the pivot chain waits, for which
tile, and what that tile was waiting for.  Usage (GPU box):  IPM_FF_TRACE_ITEMS=1 python tools/ff_trace.py [m n] [out.npz]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IPM_FF_TRACE_ITEMS", "1")
import interiorpointmethod_amd as ipm                      # noqa: E402
from interiorpointmethod_amd import _lib                   # noqa: E402
from interiorpointmethod_amd.workloads import synthetic_lp  # noqa: E402

FF_F, FF_T = 0, 1


def collect(m, n, iters=6):
    A, b, c = synthetic_lp(m, n, seed=0)
    sv = ipm.IpmSolver(A, b, c, device=0)
    sv.init_state(0.0)
    sv.iterate(iters)                                     # the trace holds the LAST launch
    lib = _lib.load()
    cnt, nit = C.c_int64(0), C.c_int32(0)
    sv._check(lib.ipm_debug_ff_trace(sv._h, None, 0, C.byref(cnt), None, C.byref(nit)))
    tr = np.zeros(cnt.value, dtype=np.int64)
    items = np.zeros(nit.value * 8, dtype=np.uint8)
    sv._check(lib.ipm_debug_ff_trace(sv._h, tr.ctypes.data_as(C.POINTER(C.c_longlong)), cnt.value, C.byref(cnt),
                                     items.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(nit)))
    nblk = sv.schedule()["blocks"] if isinstance(sv.schedule(), dict) and "blocks" in sv.schedule() else (m + 127) // 128
    sv.close()
    return tr, items.reshape(-1, 8), nit.value, nblk


def analyse(tr, items, nit, nblk, out=sys.stdout):
    it = tr[:4 * nit].reshape(nit, 4).astype(np.float64)
    ch = tr[4 * nit:].reshape(-1, 12).astype(np.float64)
    nblk = ch.shape[0]
    t0 = it[:, 0][it[:, 0] > 0].min()
    us = lambda v: (v - t0) / 100.0                       # 100 MHz clock -> microseconds
    typ, ti, tc = items[:, 0], items[:, 1].astype(int), items[:, 2].astype(int)
    j0, j1, flags, seq = items[:, 4].astype(int), items[:, 5].astype(int), items[:, 6].astype(int), items[:, 7].astype(int)
    F = typ == FF_F
    T = typ == FF_T
    print("items %d (F %d, T %d), workers %d" % (nit, F.sum(), T.sum(), int(it[:, 3].max()) + 1), file=out)
    dur_f = (it[F, 2] - it[F, 0]) / 100.0
    st = (items[F, 6].astype(int) * 256 + items[F, 4].astype(int))        # f.s0 little endian: bytes 4,5 ; s1: bytes 6,7
    s0 = items[F, 4].astype(int) + 256 * items[F, 5].astype(int)
    s1 = items[F, 6].astype(int) + 256 * items[F, 7].astype(int)
    ns = np.maximum(s1 - s0, 1)
    print("F chunks: us per BK=16 stage median %.2f (p10 %.2f p90 %.2f); formation ends at %.0f us" % (
        np.median(dur_f / ns), np.percentile(dur_f / ns, 10), np.percentile(dur_f / ns, 90), us(it[F, 2].max())), file=out)
    tw = (it[T, 1] - it[T, 0]) / 100.0
    tg = (it[T, 2] - it[T, 1]) / 100.0
    print("T items: wait sum %.0f us-CU (mean %.1f), work sum %.0f us-CU (mean %.1f); last T done at %.0f us" % (
        tw.sum(), tw.mean(), tg.sum(), tg.mean(), us(it[T, 2].max())), file=out)
    # per tile: time its formation was complete, time its last T item ended
    tile_form, tile_done, tile_last_item = {}, {}, {}
    for n_ in np.nonzero(F)[0]:
        for r in (ti[n_], ti[n_] + 1):
            if r >= tc[n_] and r < nblk:
                tile_form[(r, tc[n_])] = max(tile_form.get((r, tc[n_]), 0.0), us(it[n_, 2]))
    for n_ in np.nonzero(T)[0]:
        key = (ti[n_], tc[n_])
        if us(it[n_, 2]) >= tile_done.get(key, -1.0):
            tile_done[key] = us(it[n_, 2]); tile_last_item[key] = n_
    print("\nstep | potrf start  ready   done | panel ready  done (in: tile k+1,k formed / last T drawn, ready, done [cols]) | update ready done (in: tile k+1,k+1 ...)", file=out)
    for k in range(nblk):
        p, cp, cu = ch[k, 0:3], ch[k, 4:7], ch[k, 8:11]
        line = "%3d  | %8.0f %8.0f %8.0f |" % (k, us(p[0]), us(p[1]), us(p[2]))
        if k + 1 < nblk and cp[0] > 0:
            def tileinfo(key):
                n_ = tile_last_item.get(key)
                if n_ is None:
                    return "formed %6.0f, no T" % tile_form.get(key, -1)
                return "formed %6.0f | T#%d drawn %6.0f ready %6.0f done %6.0f [%d,%d) w%d" % (
                    tile_form.get(key, -1), n_, us(it[n_, 0]), us(it[n_, 1]), us(it[n_, 2]), j0[n_], j1[n_], int(it[n_, 3]))
            line += " %8.0f %8.0f (%s) | %8.0f %8.0f (%s)" % (us(cp[1]), us(cp[2]), tileinfo((k + 1, k)), us(cu[1]), us(cu[2]), tileinfo((k + 1, k + 1)))
        print(line, file=out)
    # chain stall accounting
    stall_p = sum(max(0.0, (ch[k, 1] - ch[k, 0]) / 100.0) for k in range(nblk))
    stall_cp = sum(max(0.0, (ch[k, 5] - ch[k, 4]) / 100.0) for k in range(nblk - 1))
    stall_cu = sum(max(0.0, (ch[k, 9] - ch[k, 8]) / 100.0) for k in range(nblk - 1))
    print("\nchain: potrf waits %.0f us, panel waits %.0f us, update waits %.0f us; chain ends at %.0f us" % (
        stall_p, stall_cp, stall_cu, us(ch[nblk - 1, 2])), file=out)
    # worker utilisation over time (100 us bins): fraction of worker time in F / T work / T wait
    end = max(it[:, 2].max(), ch[nblk - 1, 2])
    nb = int(us(end) // 100) + 1
    W = int(it[:, 3].max()) + 1
    binsF, binsT, binsW = np.zeros(nb), np.zeros(nb), np.zeros(nb)

    def add(bins, a, b_):
        a, b_ = us(a), us(b_)
        i0, i1 = int(a // 100), int(b_ // 100)
        for i in range(i0, min(i1, nb - 1) + 1):
            lo, hi = max(a, i * 100.0), min(b_, (i + 1) * 100.0)
            if hi > lo:
                bins[i] += hi - lo
    for n_ in range(nit):
        if it[n_, 0] <= 0:
            continue
        if not T[n_]:
            add(binsF, it[n_, 0], it[n_, 2])
        else:
            add(binsW, it[n_, 0], it[n_, 1]); add(binsT, it[n_, 1], it[n_, 2])
    print("\n t(us)   F%%   T%%  wait%% idle%%   chain step", file=out)
    pst = [us(ch[k, 0]) for k in range(nblk)]
    for i in range(nb):
        tot = 100.0 * W
        kk = sum(1 for v in pst if v <= (i + 1) * 100.0) - 1
        print("%6d  %4.0f %4.0f %5.0f %5.0f   %d" % (i * 100, 100 * binsF[i] / tot, 100 * binsT[i] / tot, 100 * binsW[i] / tot,
                                                    100 * (1 - (binsF[i] + binsT[i] + binsW[i]) / tot), kk), file=out)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    m, n = (int(args[0]), int(args[1])) if len(args) >= 2 else (4096, 8192)
    tr, items, nit, nblk = collect(m, n)
    if len(args) >= 3 or len(args) == 1:
        np.savez_compressed(args[-1], trace=tr, items=items, nit=nit, nblk=nblk)
    analyse(tr, items, nit, nblk)
