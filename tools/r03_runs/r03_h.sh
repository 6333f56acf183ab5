#!/bin/bash
# sparse factor with the matrix-core update of the large fronts: tests, then ms per iteration sparse / dense-tile, one LP on the GPU
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sparse_factor.py -x -q > $O/h_pytest.log 2>&1 || { tail -40 $O/h_pytest.log; exit 1; }
tail -1 $O/h_pytest.log
timeout -k 10 600 python tools/sparse_factor_check.py STOCFOR3 80BAU3B SIERRA CZPROB 25FV47 BNL2 D2Q06C GREENBEA PILOTNOV NESM WOODW TRUSS D6CUBE CYCLE GANGES SCFXM3 2>&1 | grep -v "^$" | cut -c1-200
