#!/bin/bash
# round 4 evidence run, part A: GPU test suite, default bench line, the 73-LP suite with its CPU baseline, kernel trace + PMC passes
# of the fused launch at the headline size, the default line again with the PMC file of THIS source.  tools/r04_runs/r04_final_a.sh [skip_tests]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
if [ "$1" != "skip_tests" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r04_final_pytest.log 2>&1 || { tail -40 $O/r04_final_pytest.log; exit 1; }
  tail -2 $O/r04_final_pytest.log
fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/r04_bench_first.log 2> $O/r04_bench_first.err || { tail -20 $O/r04_bench_first.err; exit 1; }
tail -1 $O/r04_bench_first.log | wc -c
timeout -k 10 300 python bench.py --workload netlib --netlib-set all > $O/r04_final_netlib_all73.log 2> $O/r04_final_netlib_all73.err || { tail -5 $O/r04_final_netlib_all73.err; exit 1; }
tail -1 $O/r04_final_netlib_all73.log | cut -c1-300
IPM_LOCKSTEP=0 timeout -k 10 300 python bench.py --workload netlib --netlib-set all --no-cpu-baseline > $O/r04_final_netlib_all73_no_lockstep.log 2>/dev/null
tail -1 $O/r04_final_netlib_all73_no_lockstep.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf $O/f_trace $O/f_pmc_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/f_trace -o p -- python3 $R/bench.py --no-netlib --no-cpu-baseline --steps 20 --warmup 2 > $O/f_trace.log 2>&1 || { tail -5 $O/f_trace.log; exit 1; }
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/f_pmc_$T -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-netlib --no-cpu-baseline > $O/f_pmc_$T.log 2>&1 || { tail -5 $O/f_pmc_$T.log; exit 1; }
done
cd $R
python tools/prof_db_stats.py $O/f_trace 16 > $O/r04_final_kernel_stats_dense.txt; head -12 $O/r04_final_kernel_stats_dense.txt | cut -c1-150
python tools/pmc_form_kernel.py --kernel form_factor_roles_kernel --out $O/r04_pmc_form_kernel_fused.json --shape 4096 8192 $O/f_pmc_* | tail -12
cp $O/r04_pmc_form_kernel_fused.json profiles/r04_pmc_form_kernel_fused.json
rm -rf $O/f_trace $O/f_pmc_*
# the default line once more, now that the PMC pass of THIS kernel source is in profiles/ (roofline.traffic)
timeout -k 10 600 python bench.py > $O/r04_final_bench.log 2> $O/r04_final_bench.err || { tail -20 $O/r04_final_bench.err; exit 1; }
tail -1 $O/r04_final_bench.log > $O/r04_final_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_final_bench.json").read())
print("final line %d bytes:" % len(open("gpurun_out/r04_final_bench.json").read()), round(d["value"], 2), "it/s; roofline", d["roofline"], d.get("cpu_baseline"))
for k in ("netlib_all", "netlib"):
    print(k, d.get(k))
PY
