#!/bin/bash
# fused formation + factorization against the serial path beyond 40 blocks: tools/ff_sizes_large.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for mn in "6144 12288" "8192 16384" "8192 9216"; do
  set -- $mn
  for ff in 0 force; do
    IPM_FUSED_FACTOR=$ff IPM_FF_MAX_NBLK=96 timeout -k 10 300 python bench.py --m $1 --n $2 --no-netlib --no-cpu-baseline --steps 10 --warmup 2 2>gpurun_out/ffl_$1_$ff.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%5d x %5d fused %s: %7.2f it/s  %.3f ms  %s  form %.3f' % ($1, $2, '$ff', d['value'], d['ms_per_step'], d['objective_check'], d['phases_ms_per_step']['form']))"
  done
done
