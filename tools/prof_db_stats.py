#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results database: python tools/prof_db_stats.py DIR_OR_DB [top]"""
import glob, os, sqlite3, sys
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
for db in dbs:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = ("select s.kernel_name, count(*), sum(k.end-k.start)/1e3, avg(k.end-k.start)/1e3, min(k.end-k.start)/1e3, "
         "max(k.end-k.start)/1e3 from %s k join %s s on k.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kt, sym))
    print(db)
    for r in c.execute(q).fetchall()[:top]:
        print("  %-56s n=%6d tot=%10.1fus avg=%9.2f min=%8.2f max=%9.2f" % (r[0][:56], r[1], r[2], r[3], r[4], r[5]))
