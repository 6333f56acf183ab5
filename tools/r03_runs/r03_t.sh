#!/bin/bash
# ragged groups of the grouped triangular solves: tests, lone LPs, dense sizes whose block count is no multiple of 8, the suite
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_factor.py -x -q > $O/t_pytest.log 2>&1 || { tail -30 $O/t_pytest.log; exit 1; }
tail -1 $O/t_pytest.log
for T in 0 1; do
  for NM in BNL2 D2Q06C FINNIS; do IPM_RAGGED_GROUPS=$T python3 tools/ss_timeline.py $NM 40 2>&1 | tail -1 | sed "s/^/IPM_RAGGED_GROUPS=$T single-stream /"; done
  IPM_RAGGED_GROUPS=$T timeout -k 10 300 python tools/sparse_factor_check.py --no-sparse PILOT87 MAROS-R7 BNL2 PILOT 2>&1 | grep -v "^$\|amdgpu.ids" | awk -v t=$T '{print "IPM_RAGGED_GROUPS=" t " lone", $1, $3, $4, $(NF-2), $(NF-1)}'
  for mn in "3584 7168" "4480 8960"; do
    set -- $mn
    IPM_RAGGED_GROUPS=$T timeout -k 10 200 python bench.py --m $1 --n $2 --no-netlib --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('IPM_RAGGED_GROUPS=$T dense %5d x %5d: %7.2f it/s  %.3f ms  %s' % ($1, $2, d['value'], d['ms_per_step'], d['objective_check']))"
  done
done
for T in 0 1 0 1; do
  IPM_RAGGED_GROUPS=$T timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/t_netlib_$T.json 2> $O/t_netlib_$T.err || { tail -5 $O/t_netlib_$T.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/t_netlib_$T.json').read().strip().splitlines()[-1]); s=d['summary']
print('IPM_RAGGED_GROUPS=$T: %.2f LPs/s wall %.3f converged %d iterations %d' % (d['value'], d['wall_seconds'], s['converged'], s['total_iterations']))"
done
