#!/bin/bash
# fused formation + factorization against the serial path over sizes: tools/ff_sizes.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for mn in "2048 4096" "2560 5120" "3072 6144" "3584 7168" "4096 8192" "4096 16384" "4096 4608" "5120 10240"; do
  set -- $mn
  for ff in 0 1; do
    IPM_FUSED_FACTOR=$ff IPM_FF_MAX_NBLK=64 timeout -k 10 200 python bench.py --m $1 --n $2 --no-netlib --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%5d x %5d fused %s: %7.2f it/s  %.3f ms  %s' % ($1, $2, '$ff', d['value'], d['ms_per_step'], d['objective_check']))"
  done
done
