#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/spf_sweep.log
timeout -k 10 300 python tools/sparse_factor_check.py --kernel --no-dense CZPROB STOCFOR2 2>&1 | cut -c1-12,200-330 >> gpurun_out/spf_sweep.log || exit 1
for cfg in "64" "256"; do
  echo "== THREADS=$cfg" >> gpurun_out/spf_sweep.log
  IPM_SP_THREADS=$cfg timeout -k 10 300 python tools/sparse_factor_check.py --no-dense STOCFOR3 STOCFOR2 SIERRA 80BAU3B CZPROB SCTAP3 SHELL GANGES 2>&1 | grep sparse | awk '{print $1, $3, $4, $(NF-3), $(NF-2), $(NF-1)}' >> gpurun_out/spf_sweep.log || exit 1
done
cat gpurun_out/spf_sweep.log
