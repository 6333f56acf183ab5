#!/bin/bash
# kernel statistics of one Netlib solve: tools/prof_one.sh NAME [dense|sparse|auto]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
NM=${1:-DEGEN3}
export IPM_FACTOR=${2:-auto}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$NM
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$NM -o p -- python3 $R/tools/solve_one.py $NM > $R/gpurun_out/prof_$NM.log 2>&1 || { tail -5 $R/gpurun_out/prof_$NM.log; exit 1; }
cd $R && tail -2 gpurun_out/prof_$NM.log && python tools/prof_db_stats.py gpurun_out/prof_$NM 16 && rm -f gpurun_out/prof_$NM/*.db
