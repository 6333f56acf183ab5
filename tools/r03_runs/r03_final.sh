#!/bin/bash
# round 3 evidence run: GPU test suite, default bench line, kernel trace + PMC passes of the fused launch at the headline
# size, its in-kernel cycle profile, the sizes sweep, config 5.   tools/r03_runs/r03_final.sh [skip_tests]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
if [ "$1" != "skip_tests" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/f_pytest.log 2>&1 || { tail -40 $O/f_pytest.log; exit 1; }
  tail -2 $O/f_pytest.log
fi
timeout -k 10 600 python bench.py > $O/f_bench.json 2> $O/f_bench.err || { tail -20 $O/f_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/f_bench.json").read().strip().splitlines()[-1])
print("dense", round(d["value"], 2), "it/s; roofline", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"][:40], d["objective_check"], {k: round(v, 3) for k, v in d["phases_ms_per_step"].items() if isinstance(v, float)})
for k in ("netlib_all", "netlib"):
    n = d[k]
    print(k, round(n["value"], 2), "LPs/s wall", round(n["wall_seconds"], 3), {q: n["summary"][q] for q in ("n", "converged", "timeouts_recovered", "serial_launches")})
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $O/f_trace $O/f_pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/f_trace -o p -- python3 $R/bench.py --no-netlib --no-cpu-baseline --steps 20 --warmup 2 > $O/f_trace.log 2>&1 || { tail -5 $O/f_trace.log; exit 1; }
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/f_pmc_$T -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-netlib --no-cpu-baseline > $O/f_pmc_$T.log 2>&1 || { tail -5 $O/f_pmc_$T.log; exit 1; }
done
cd $R
python tools/prof_db_stats.py $O/f_trace 16 > $O/f_kernel_stats.txt; head -12 $O/f_kernel_stats.txt | cut -c1-150
python tools/pmc_form_kernel.py --kernel form_factor_kernel --out $O/f_pmc_form_kernel.json --shape 4096 8192 $O/f_pmc_* | tail -12
cp $O/f_pmc_form_kernel.json profiles/r03_pmc_form_kernel_fused.json
python - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/f_trace/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]; sym = [t for t in tabs if "kernel_symbol" in t][0]
rows = c.execute("select s.kernel_name, k.start, k.end from %s k join %s s on k.kernel_id=s.id order by k.start" % (kt, sym)).fetchall()
ff = [i for i, r in enumerate(rows) if "form_factor_kernel" in r[0]]
if len(ff) > 12:
    i0 = ff[10]; t0 = rows[i0][1]; t1 = rows[ff[11]][1]
    out = ["one iteration of the fused path under rocprofv3 --kernel-trace: start (us, relative to the worker launch), duration (us), kernel"]
    for r in rows[i0 - 4:]:
        if r[1] >= t1: break
        out.append("%9.1f %9.1f  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0][:70]))
    open("gpurun_out/f_timeline.txt", "w").write("\n".join(out) + "\n")
PY
rm -f $O/f_trace/*.db $O/f_trace/*/*.db
IPM_FF_PROF=1 timeout -k 10 200 python bench.py --no-netlib --no-cpu-baseline --steps 20 --warmup 2 2>&1 >/dev/null | grep -A10 "ff prof" > $O/f_cycle_profile.txt; cat $O/f_cycle_profile.txt
bash tools/ff_sizes.sh > $O/f_sizes.txt 2>&1; cat $O/f_sizes.txt
timeout -k 10 400 python bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib --no-cpu-baseline > $O/f_dense16k.json 2> $O/f_dense16k.err || { tail -5 $O/f_dense16k.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/f_dense16k.json').read().strip().splitlines()[-1]); print('16k', d['value'], d['roofline']['frac'], d['roofline']['traffic_source'], d['objective_check'])"
# the default line once more, now that the PMC pass of THIS kernel source is in profiles/ (roofline.traffic)
timeout -k 10 600 python bench.py > $O/f_bench_final.json 2> $O/f_bench_final.err || { tail -20 $O/f_bench_final.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/f_bench_final.json').read().strip().splitlines()[-1]); print('final', round(d['value'],2), d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'], d['objective_check'], round(d['netlib_all']['value'],2), round(d['netlib']['value'],2))"
