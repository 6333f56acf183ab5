#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 10 900 python tools/sparse_factor_check.py GFRD-PNC SCFXM2 SCFXM3 SCTAP2 NESM GROW15 GANGES BNL1 ETAMACRO > gpurun_out/spf_check.log 2>&1
echo "rc=$?" >> gpurun_out/spf_check.log
grep -v amdgpu.ids gpurun_out/spf_check.log | awk '{print $1, $2, $3, $4, $(NF-4), $(NF-3), $(NF-2), $(NF-1)}'
