#!/bin/bash
# adat_sparse_kernel with the per-nonzero metadata fetched 256 at a time: tests, lone LPs
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/y_pytest.log 2>&1 || { tail -30 $O/y_pytest.log; exit 1; }
tail -1 $O/y_pytest.log
timeout -k 10 400 python tools/sparse_factor_check.py --no-sparse PILOT87 MAROS-R7 BNL2 PILOT DFL001 QAP15 2>&1 | grep -v "^$\|amdgpu.ids" | awk '{print "lone", $1, $3, $4, $(NF-2), $(NF-1)}'
for NM in BNL2 D2Q06C; do python3 tools/ss_timeline.py $NM 40 2>&1 | tail -1; done
