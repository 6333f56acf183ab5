#!/bin/bash
# 73-LP suite under the lockstep batches, a few settings each (GPU box): tools/sweep_ls.sh > gpurun_out/sweep_ls.txt
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --workload netlib --no-cpu-baseline $SET 2>gpurun_out/sweep_ls_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %.2f LPs/s wall %.3f s converged %d slowest %s' % (d['value'], d['wall_seconds'], d['summary']['converged'], d.get('slowest_lp')))"; grep "^\[batch\]\|^\[lockstep\] batch" gpurun_out/sweep_ls_err.txt; }
for rep in 1 2; do
SET="--netlib-set all" run IPM_LS_DEBUG=1
SET="--netlib-set parity" run IPM_LS_DEBUG=1
SET="--netlib-set all" run IPM_LOCKSTEP=0
SET="--netlib-set parity" run IPM_LOCKSTEP=0
done
