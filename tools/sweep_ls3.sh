#!/bin/bash
# 73-LP suite: aligned merge vs leader rule, class limits (GPU box): tools/sweep_ls3.sh > gpurun_out/sweep_ls3.txt
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --workload netlib --no-cpu-baseline $SET 2>gpurun_out/sweep_ls_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %.2f LPs/s wall %.3f s converged %d slowest %s' % (d['value'], d['wall_seconds'], d['summary']['converged'], d.get('slowest_lp')))"; grep "^\[batch\]\|^\[lockstep\] batch" gpurun_out/sweep_ls_err.txt; grep "merged schedule" gpurun_out/sweep_ls_err.txt | sort -t' ' -k2 -n | tail -2 | cut -c1-200; }
for rep in 1 2; do
SET="--netlib-set all" run IPM_LS_DEBUG=1
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LS_MERGE=leader
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSES=3500
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSES=1000,2200,3500
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSES=2200,3500 IPM_LOCKSTEP_CLASSIC_THREADS=2
SET="--netlib-set parity" run IPM_LS_DEBUG=1
SET="--netlib-set parity" run IPM_LOCKSTEP=0
done
