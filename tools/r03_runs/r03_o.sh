#!/bin/bash
# kernel timeline of one iteration of a mid-size LP on a single-stream handle
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
for NM in DEGEN3 BNL2 FINNIS; do
  python3 tools/ss_timeline.py $NM 24 2>&1 | tail -1
  cd /tmp && export TMPDIR=/tmp; rm -rf $O/o_trace_$NM
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/o_trace_$NM -o p -- python3 $R/tools/ss_timeline.py $NM 24 > $O/o_trace_$NM.log 2>&1 || { tail -5 $O/o_trace_$NM.log; exit 1; }
  cd $R
  python3 - $NM <<'PY'
import glob, sqlite3, sys
nm = sys.argv[1]
db = glob.glob("gpurun_out/o_trace_%s/**/*.db" % nm, recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute("select s.kernel_name, k.start, k.end from %s k join %s s on k.kernel_id=s.id order by k.start" % (kt, sym)).fetchall()
pp = [i for i, r in enumerate(rows) if "prepare_kernel" in r[0]]
i0, i1 = pp[12], pp[13]
t0 = rows[i0][1]
out = ["%s, single-stream handle: one iteration, start (us), duration (us), gap to the previous kernel's end (us), kernel" % nm]
prev_end = rows[i0 - 1][2]
busy = 0.0
import collections
agg = collections.OrderedDict()
for r in rows[i0:i1]:
    out.append("%9.1f %8.1f %6.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3, r[0][:80]))
    key = r[0][:60]
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += (r[2] - r[1]) / 1e3; a[2] += max(0.0, (r[1] - prev_end) / 1e3)
    busy += (r[2] - r[1]) / 1e3
    prev_end = r[2]
tot = (rows[i1][1] - t0) / 1e3
out.append("iteration %.1f us, kernels %d, busy %.1f us, gaps %.1f us" % (tot, i1 - i0, busy, tot - busy))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append("  %4d x  busy %8.1f  gaps before %7.1f  %s" % (a[0], a[1], a[2], k))
open("gpurun_out/o_timeline_%s.txt" % nm, "w").write("\n".join(out) + "\n")
print("\n".join(out[-22:]))
PY
  rm -rf $O/o_trace_$NM
done
