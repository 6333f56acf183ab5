#!/bin/bash
# end-of-round verification: GPU test suite, default bench line, full Netlib suite, kernel statistics
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1 || { tail -30 gpurun_out/final_pytest.log; exit 1; }
tail -2 gpurun_out/final_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -5 gpurun_out/final_bench.err; exit 1; }
timeout -k 10 300 python bench.py --workload netlib --netlib-set all > gpurun_out/final_netlib_all.json 2> gpurun_out/final_netlib_all.err || { tail -5 gpurun_out/final_netlib_all.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof_dense -o p -- python3 $R/bench.py --no-netlib --no-cpu-baseline > $R/gpurun_out/final_prof_dense.log 2>&1 || { tail -5 $R/gpurun_out/final_prof_dense.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof_sparse -o p -- python3 $R/tools/sparse_factor_check.py --no-dense STOCFOR3 > $R/gpurun_out/final_prof_sparse.log 2>&1 || { tail -5 $R/gpurun_out/final_prof_sparse.log; exit 1; }
cd $R
python tools/prof_db_stats.py gpurun_out/final_prof_dense 14 > gpurun_out/final_kernel_stats_dense.txt
python tools/prof_db_stats.py gpurun_out/final_prof_sparse 14 > gpurun_out/final_kernel_stats_sparse.txt
find gpurun_out/final_prof_dense gpurun_out/final_prof_sparse -name "*stats*.csv" | head
rm -f gpurun_out/final_prof_dense/*.db gpurun_out/final_prof_sparse/*.db
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print("dense", d["value"], d["roofline"]["frac"], "netlib parity", d["netlib"]["value"], d["netlib"]["wall_seconds"])
n=json.loads(open("gpurun_out/final_netlib_all.json").read().strip().splitlines()[-1])
print("netlib all", n["value"], n["wall_seconds"], n["summary"])
PY
# larger dense sizes (MFMA-utilisation evidence, BASELINE config 5)
timeout -k 10 300 python bench.py --m 8192 --n 16384 --steps 10 --warmup 2 --no-netlib --no-cpu-baseline > gpurun_out/final_dense8k.json 2> gpurun_out/final_dense8k.err || { tail -5 gpurun_out/final_dense8k.err; exit 1; }
timeout -k 10 400 python bench.py --m 16384 --n 32768 --steps 5 --warmup 1 --no-netlib --no-cpu-baseline > gpurun_out/final_dense16k.json 2> gpurun_out/final_dense16k.err || { tail -5 gpurun_out/final_dense16k.err; exit 1; }
python - <<'PY'
import json
for f in ("final_dense8k", "final_dense16k"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["roofline"]["frac"], d.get("whole_iteration", {}).get("frac_of_fp64_mfma_peak"))
PY
