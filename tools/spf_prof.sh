#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for nm in GROW22 GREENBEA STOCFOR3; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/spf_prof_$nm -o p -- python3 $R/tools/sparse_factor_check.py --no-dense $nm > $R/gpurun_out/spf_prof_$nm.log 2>&1 || exit 1
done
echo ok
