#!/usr/bin/env python3
"""Where a multi-stream run's time goes, from a rocprofv3 results database: python tools/ls_gaps.py DIR_OR_DB
Per stream (= hardware queue client): launches, kernel time, gaps between consecutive kernels; then kernel duration and gap as a
function of how many OTHER streams had a kernel executing at that moment (is it the kernels or the hand-over that inflates?)."""
import glob, os, sqlite3, sys
import numpy as np
path = sys.argv[1]
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
for db in dbs:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kt)]
    print(db, cols)
    key = "stream_id" if "stream_id" in cols else "queue_id"
    rows = np.array(c.execute("select %s, queue_id, start, end from %s order by start" % (key, kt)).fetchall(), dtype=np.int64)
    t0 = rows[:, 2].min()
    rows[:, 2:] -= t0
    span = rows[:, 3].max() / 1e3
    print("launches %d, span %.1f ms, kernel time %.1f ms" % (len(rows), span / 1e3, (rows[:, 3] - rows[:, 2]).sum() / 1e6))
    # concurrency at each kernel start: number of kernels of OTHER streams executing
    ev = np.concatenate([np.stack([rows[:, 2], np.ones(len(rows), np.int64)], 1), np.stack([rows[:, 3], -np.ones(len(rows), np.int64)], 1)])
    ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
    lvl = np.cumsum(ev[:, 1])
    def level_at(t):
        return lvl[np.searchsorted(ev[:, 0], t, side="right") - 1]
    for s in np.unique(rows[:, 0]):
        r = rows[rows[:, 0] == s]
        if len(r) < 200:
            continue
        d = (r[:, 3] - r[:, 2]) / 1e3
        g = (r[1:, 2] - r[:-1, 3]) / 1e3
        gs = g[(g > -50) & (g < 200)]
        q = np.unique(r[:, 1])
        print("stream %d (queues %s): %6d launches over %.1f ms (%.1f ... %.1f), kernels %.1f ms (median %.1f us), gaps<200us %.1f ms (median %.2f, p90 %.2f), gaps>=200us %.1f ms"
              % (s, q.tolist(), len(r), (r[-1, 3] - r[0, 2]) / 1e6, r[0, 2] / 1e6, r[-1, 3] / 1e6, d.sum() / 1e3, np.median(d), gs.sum() / 1e3, np.median(gs), np.percentile(gs, 90), g[g >= 200].sum() / 1e3))
        lv = level_at(r[1:, 2]) - 1
        for k in range(0, 8):
            m = (lv == k) & (g > -50) & (g < 200)
            if m.sum() > 50:
                print("     %d other kernels executing: %6d launches, gap median %.2f us mean %.2f, kernel median %.1f us mean %.1f"
                      % (k, m.sum(), np.median(g[m]), g[m].mean(), np.median(d[1:][m]), d[1:][m].mean()))
