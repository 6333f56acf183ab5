#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 300 python tools/sparse_factor_check.py --kernel --no-dense BANDM CZPROB GROW22 GREENBEA > gpurun_out/spf_kernel.log 2>&1
echo "rc=$?" >> gpurun_out/spf_kernel.log
cut -c1-330 gpurun_out/spf_kernel.log | tail -9
timeout -k 10 900 python tools/sparse_factor_check.py --no-dense SC205 25FV47 SIERRA STOCFOR2 BNL2 80BAU3B STOCFOR3 D2Q06C SCTAP3 SHELL GANGES NESM SCFXM3 > gpurun_out/spf_check.log 2>&1
echo "rc=$?" >> gpurun_out/spf_check.log
cut -c1-175 gpurun_out/spf_check.log | tail -40
