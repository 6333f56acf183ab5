#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sparse_factor.py "tests/test_gpu_parity.py::test_netlib_suite_batched_config4" "tests/test_gpu_parity.py::test_netlib_parity" -m gpu -x -q > gpurun_out/pytest_spf.log 2>&1 || { tail -30 gpurun_out/pytest_spf.log; exit 1; }
tail -2 gpurun_out/pytest_spf.log
timeout -k 10 600 python tools/sparse_factor_check.py --no-dense STOCFOR3 SIERRA STOCFOR2 CZPROB SCTAP3 SHELL 80BAU3B GANGES SCFXM3 NESM GREENBEA > gpurun_out/spf_check.log 2>&1
grep -v amdgpu.ids gpurun_out/spf_check.log | awk '{print $1, $2, $3, $4, $(NF-4), $(NF-3), $(NF-2), $(NF-1)}'
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/spf_prof_STOCFOR3
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/spf_prof_STOCFOR3 -o p -- python3 $R/tools/sparse_factor_check.py --no-dense STOCFOR3 > $R/gpurun_out/spf_prof_STOCFOR3.log 2>&1
cd $R && python tools/prof_db_stats.py gpurun_out/spf_prof_STOCFOR3 4; rm -f gpurun_out/spf_prof_STOCFOR3/*.db
