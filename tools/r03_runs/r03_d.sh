#!/bin/bash
# bulk trailing-update kernel: bitwise test, then A/B at the sizes whose factorization is throughput bound
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bulk_update or two_level or cholesky_and_solve" > $O/d_pytest.log 2>&1 || { tail -40 $O/d_pytest.log; exit 1; }
tail -1 $O/d_pytest.log
for mn in "4096 8192" "8192 16384" "16384 32768"; do
  set -- $mn
  for bv in 7 0; do
    steps=20; [ $1 -ge 16384 ] && steps=6
    IPM_FUSED_FACTOR=0 IPM_BULK_VARIANT=$bv timeout -k 10 400 python bench.py --m $1 --n $2 --no-netlib --no-cpu-baseline --steps $steps --warmup 2 2>$O/d_$1_$bv.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phases_ms_per_step']; print('%5d x %5d bulk variant %s: %7.3f it/s  %.3f ms  %s  form %.3f factor %.3f' % ($1, $2, '$bv', d['value'], d['ms_per_step'], d['objective_check'], p['form'], p['factor']))"
  done
done
