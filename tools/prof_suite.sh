#!/bin/bash
# kernel statistics of a batched Netlib run (8 LPs in flight): tools/prof_suite.sh [parity|all]  (rocprofv3 crashed on "all" once: ~2e6 dispatches)
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_suite
timeout -k 10 600 rocprofv3 --kernel-trace -d /tmp/prof_suite -o p -- python3 $R/bench.py --workload netlib --netlib-set ${1:-parity} --no-cpu-baseline > $R/gpurun_out/prof_suite.log 2>&1 || { tail -5 $R/gpurun_out/prof_suite.log; exit 1; }
cd $R && python tools/prof_db_stats.py /tmp/prof_suite 24 > gpurun_out/prof_suite_stats.txt; cat gpurun_out/prof_suite_stats.txt
