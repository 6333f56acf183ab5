#!/bin/bash
# round 3, fused formation + factorization: correctness tests, then timing and a kernel trace of the headline size
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused_factor.py -x -q > $O/b_pytest.log 2>&1 || { tail -40 $O/b_pytest.log; exit 1; }
tail -2 $O/b_pytest.log
for FF in 0 1; do
  IPM_FUSED_FACTOR=$FF timeout -k 10 300 python bench.py --no-netlib --no-cpu-baseline --steps 40 > $O/b_bench_ff$FF.json 2> $O/b_bench_ff$FF.err || { tail -5 $O/b_bench_ff$FF.err; exit 1; }
  python - $FF <<'PY'
import json,sys
d=json.loads(open("gpurun_out/b_bench_ff%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("FF",sys.argv[1],"it/s",round(d["value"],2),"ms",round(d["ms_per_step"],3),d["objective_check"],"form(ev) ms",round(d["phases_ms_per_step"]["form"],3))
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/b_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/b_trace -o p -- python3 $R/bench.py --no-netlib --no-cpu-baseline --steps 20 --warmup 2 > $O/b_trace.log 2>&1 || { tail -5 $O/b_trace.log; exit 1; }
cd $R && python tools/prof_db_stats.py $O/b_trace 14 > $O/b_kernel_stats.txt; cat $O/b_kernel_stats.txt | cut -c1-150
python - <<'PY'
# one iteration's timeline of the chain kernels and the worker launch (start offsets in us)
import glob, sqlite3
db = glob.glob("gpurun_out/b_trace/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute("select s.kernel_name, k.start, k.end from %s k join %s s on k.kernel_id=s.id order by k.start" % (kt, sym)).fetchall()
ff = [i for i, r in enumerate(rows) if "form_factor_kernel" in r[0]]
if len(ff) > 12:
    i0 = ff[10]; t0 = rows[i0][1]; t1 = rows[ff[11]][1]
    out = []
    for r in rows[i0 - 3:]:
        if r[1] >= t1: break
        out.append("%9.1f %9.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:60]))
    open("gpurun_out/b_timeline.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out[:12])); print("..."); print("\n".join(out[-14:]))
PY
rm -f $O/b_trace/*.db $O/b_trace/*/*.db
