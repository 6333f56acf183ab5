#!/bin/bash
# kernel statistics of one lockstep batch of the named LPs: tools/prof_probe.sh NAME ...
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_probe
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_probe -o p -- python3 $R/tools/ls_probe.py "$@" > $R/gpurun_out/prof_probe.log 2>&1 || { tail -5 $R/gpurun_out/prof_probe.log; exit 1; }
cd $R && grep "batch of" gpurun_out/prof_probe.log && python tools/prof_db_stats.py gpurun_out/prof_probe 30 && python tools/ls_gaps.py gpurun_out/prof_probe && rm -rf gpurun_out/prof_probe
