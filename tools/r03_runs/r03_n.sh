#!/bin/bash
# look-ahead handles (one LP on the GPU): narrow-tile kernels on the bulk stream for the last trailing blocks
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
IPM_LA_SMALL_TILES=16 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cholesky_and_solve or two_level or netlib_parity or bulk_update or drop_in or dense_kat" > $O/n_pytest.log 2>&1 || { tail -30 $O/n_pytest.log; exit 1; }
tail -1 $O/n_pytest.log
for T in 0 8 16 32; do
  echo "== IPM_LA_SMALL_TILES=$T"
  IPM_LA_SMALL_TILES=$T timeout -k 10 300 python tools/sparse_factor_check.py --no-sparse DEGEN3 BNL2 25FV47 PILOT87 MAROS-R7 D2Q06C PILOTNOV FINNIS SCFXM3 2>&1 | grep -v "^$\|amdgpu.ids" | awk '{print $1, $2, $3, $4, $(NF-2), $(NF-1)}'
done
