#!/bin/bash
# fused formation + factorization against the serial path over sizes: tools/ff_sizes.sh [small|large]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
if [ "$1" = "large" ]; then SIZES=("6144 12288" "8192 16384" "8192 9216" "10240 20480"); ST=10; else
SIZES=("1536 3072" "2048 4096" "2560 5120" "3072 6144" "3584 7168" "4096 8192" "4096 16384" "4096 32768" "4096 4608" "5120 10240"); ST=20; fi
for mn in "${SIZES[@]}"; do
  set -- $mn
  for ff in 0 force; do
    IPM_FUSED_FACTOR=$ff IPM_FF_MAX_NBLK=96 timeout -k 10 300 python bench.py --m $1 --n $2 --no-netlib --no-cpu-baseline --steps $ST --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%5d x %5d fused %-5s: %7.2f it/s  %.3f ms  launch %.3f ms  %s' % ($1, $2, '$ff', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['objective_check']))"
  done
done
