#!/bin/bash
# kernel statistics of the batched Netlib suite (lockstep batch): tools/prof_suite.sh
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_suite
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_suite -o p -- python3 $R/bench.py --workload netlib --no-cpu-baseline "$@" > $R/gpurun_out/prof_suite.log 2>&1 || { tail -5 $R/gpurun_out/prof_suite.log; exit 1; }
cd $R && tail -1 gpurun_out/prof_suite.log | cut -c1-160 && python tools/prof_db_stats.py gpurun_out/prof_suite 40 && rm -f gpurun_out/prof_suite/*.db gpurun_out/prof_suite/*/*.db
