#!/bin/bash
# sparse factor: forward substitution fused into the factorization (4 tree walks per iteration) -- tests, A/B per LP, suite
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_sparse_factor.py -x -q > $O/j_pytest.log 2>&1 || { tail -40 $O/j_pytest.log; exit 1; }
tail -1 $O/j_pytest.log
for FW in 0 1; do
  for SC in 0 1; do
    echo "== IPM_SP_FUSE_FWD=$FW IPM_SP_SC1=$SC"
    IPM_SP_FUSE_FWD=$FW IPM_SP_SC1=$SC timeout -k 10 300 python tools/sparse_factor_check.py --no-dense STOCFOR3 80BAU3B SIERRA CZPROB STOCFOR2 SCTAP3 SHELL GANGES 2>&1 | grep -v "^$\|amdgpu.ids" | awk '{print $1, $2, $3, $4, $(NF-4), $(NF-3), $(NF-2)}'
  done
done
for FW in 0 1; do
  IPM_SP_FUSE_FWD=$FW timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/j_netlib_fw$FW.json 2> $O/j_netlib_fw$FW.err || { tail -5 $O/j_netlib_fw$FW.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/j_netlib_fw$FW.json').read().strip().splitlines()[-1]); s=d['summary']
print('suite FUSE_FWD=$FW: %.2f LPs/s wall %.3f converged %d iterations %d' % (d['value'], d['wall_seconds'], s['converged'], s['total_iterations']))"
done
