#!/bin/bash
# kernel trace of the batched Netlib suite + the per-stream gap analysis: tools/prof_suite_gaps.sh [env assignments ...]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
for a in "$@"; do export "$a"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_gaps
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_gaps -o p -- python3 $R/bench.py --workload netlib --no-cpu-baseline > $R/gpurun_out/prof_gaps.log 2>&1 || { tail -5 $R/gpurun_out/prof_gaps.log; exit 1; }
cd $R && tail -1 gpurun_out/prof_gaps.log | cut -c1-200 && python tools/ls_gaps.py gpurun_out/prof_gaps && rm -rf gpurun_out/prof_gaps
