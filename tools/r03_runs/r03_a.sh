#!/bin/bash
# round 3, first GPU call: the GPU test suite, the default bench line (now with netlib_all), BASELINE config 5's profiler
# evidence (kernel trace + PMC passes at 16384 x 32768) and ONE profiled run of the batched 73-LP suite with the process'
# address map dumped (to symbolise the round-2 abort if it shows again).   tools/r03_runs/r03_a.sh [skip_tests]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
O=$R/gpurun_out
mkdir -p $O
if [ "$1" != "skip_tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/a_pytest.log 2>&1 || { tail -40 $O/a_pytest.log; exit 1; }
  tail -2 $O/a_pytest.log
fi
timeout -k 10 600 python bench.py > $O/a_bench.json 2> $O/a_bench.err || { tail -20 $O/a_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/a_bench.json").read().strip().splitlines()[-1])
print("dense", round(d["value"], 2), "frac", round(d["roofline"]["frac"], 4), d["objective_check"], d["phases_ms_per_step"])
for k in ("netlib_all", "netlib"):
    n = d[k]
    print(k, round(n["value"], 2), "LPs/s wall", round(n["wall_seconds"], 3), {q: n["summary"][q] for q in ("n", "converged", "timeouts_recovered", "serial_launches")},
          "largest", n["roofline"]["latency_floor"]["largest_lp"], round(n["roofline"]["latency_floor"]["largest_lp_seconds"], 3),
          "cpu", round(n.get("cpu_baseline", {}).get("value", 0), 2), len(n.get("cpu_baseline", {}).get("sample_names", [])))
slow = sorted(d["netlib_all"]["per_lp"].items(), key=lambda kv: -kv[1]["s"])[:12]
print("slowest:", [(k, v["s"], v["setup_s"], v["solve_s"], v["it"]) for k, v in slow])
PY
cd /tmp && export TMPDIR=/tmp
# --- config 5: kernel trace + stats, then one PMC pass per counter set (never combined with a trace domain other than kernel-trace)
rm -rf $O/p16k_trace $O/p16k_pmc_*
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/p16k_trace -o p -- python3 $R/bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib --no-cpu-baseline > $O/p16k_trace.log 2>&1 || { tail -5 $O/p16k_trace.log; exit 1; }
tail -1 $O/p16k_trace.log | cut -c1-600
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p16k_pmc_$T -o p -- python3 $R/bench.py --m 16384 --n 32768 --steps 3 --warmup 1 --no-netlib --no-cpu-baseline > $O/p16k_pmc_$T.log 2>&1 || { tail -5 $O/p16k_pmc_$T.log; exit 1; }
  echo "pmc $T done"
done
cd $R
python tools/prof_db_stats.py $O/p16k_trace 16 > $O/p16k_kernel_stats.txt; head -20 $O/p16k_kernel_stats.txt
python tools/pmc_form_kernel.py --out $O/p16k_pmc_form_kernel.json --shape 16384 32768 $O/p16k_pmc_* | tail -25
rm -f $O/p16k_trace/*.db $O/p16k_trace/*/*.db
# --- the batched 73-LP suite under the profiler, once: 2 LPs in flight first (kernel statistics for profiles/), then 8
cd /tmp
rm -rf /tmp/prof_suite2 /tmp/prof_suite8
IPM_DUMP_MAPS=$O/suite_w2_maps.txt timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/prof_suite2 -o p -- python3 $R/bench.py --workload netlib --netlib-set all --workers 2 --no-cpu-baseline > $O/prof_suite_w2.log 2>&1; echo "suite w2 rc $?"
(cd $R && python tools/prof_db_stats.py /tmp/prof_suite2 30 > $O/prof_suite_w2_stats.txt 2>&1; head -12 $O/prof_suite_w2_stats.txt)
IPM_DUMP_MAPS=$O/suite_w8_maps.txt timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/prof_suite8 -o p -- python3 $R/bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/prof_suite_w8.log 2>&1; echo "suite w8 rc $?"
(cd $R && python tools/prof_db_stats.py /tmp/prof_suite8 30 > $O/prof_suite_w8_stats.txt 2>&1; head -12 $O/prof_suite_w8_stats.txt)
tail -3 $O/prof_suite_w8.log | cut -c1-300
exit 0
