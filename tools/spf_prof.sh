#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for div in 1 1000000; do
  rm -rf $R/gpurun_out/spf_prof_div$div
  IPM_SP_TASK_DIV=$div timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/spf_prof_div$div -o p -- python3 $R/tools/sparse_factor_check.py --no-dense STOCFOR3 > $R/gpurun_out/spf_prof_div$div.log 2>&1 || exit 1
  (cd $R && python tools/prof_db_stats.py gpurun_out/spf_prof_div$div 3 && grep "sparse status" gpurun_out/spf_prof_div$div.log | awk '{print $(NF-3), $(NF-2), $(NF-1)}')
  rm -f $R/gpurun_out/spf_prof_div$div/*.db
done
