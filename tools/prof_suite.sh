#!/bin/bash
# kernel statistics of a batched Netlib run: tools/prof_suite.sh [parity|all] [workers]
# Default TWO LPs in flight: with eight, rocprofv3's AQL packet interceptor (librocprofiler-sdk.so.1.1.0 +0x1e72fb, below
# hipLaunchKernel) walks off the end of its packet array and the process dies with SIGSEGV -- resolved from a dumped
# /proc/self/maps in profiles/r03_prof_suite_abort_symbolised.txt; the same run without the tool is clean.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_suite
IPM_DUMP_MAPS=$R/gpurun_out/prof_suite_maps.txt timeout -k 10 600 rocprofv3 --kernel-trace -d /tmp/prof_suite -o p -- python3 $R/bench.py --workload netlib --netlib-set ${1:-all} --workers ${2:-2} --no-cpu-baseline > $R/gpurun_out/prof_suite.log 2>&1 || { tail -5 $R/gpurun_out/prof_suite.log; exit 1; }
cd $R && python tools/prof_db_stats.py /tmp/prof_suite 30 > gpurun_out/prof_suite_stats.txt; cat gpurun_out/prof_suite_stats.txt
