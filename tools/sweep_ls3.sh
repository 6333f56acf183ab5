#!/bin/bash
# 73-LP suite: class limits of the lockstep batches (GPU box): tools/sweep_ls3.sh > gpurun_out/sweep_ls3.txt
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --workload netlib --no-cpu-baseline $SET 2>gpurun_out/sweep_ls_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %.2f LPs/s wall %.3f s converged %d slowest %s' % (d['value'], d['wall_seconds'], d['summary']['converged'], d.get('slowest_lp')))"; grep "^\[batch\]\|^\[lockstep\] batch" gpurun_out/sweep_ls_err.txt | cut -c1-150; }
for rep in 1 2; do
SET="--netlib-set all" run IPM_LS_DEBUG=1
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSES=2200,4000 IPM_LOCKSTEP_DENSE_ROWS=4000
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSES=2500,4000 IPM_LOCKSTEP_DENSE_ROWS=4000
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSES=1800,4000 IPM_LOCKSTEP_DENSE_ROWS=4000
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSES=2200,4000 IPM_LOCKSTEP_DENSE_ROWS=6000
done
