#!/bin/bash
# 73-LP suite under the lockstep batches with the streams created once: hardware queues x streams (GPU box):
#   tools/sweep_ls3.sh > gpurun_out/sweep_ls5.txt      (-> profiles/r04_netlib_hw_queues_x_streams_sweep.txt)
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --workload netlib --no-cpu-baseline --netlib-set all 2>gpurun_out/sweep_ls_err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %.2f LPs/s wall %.3f s converged %d slowest %s it %d' % (d['value'], d['wall_seconds'], d['summary']['converged'], d.get('slowest_lp'), d['summary']['total_iterations']))"; grep "^\[batch\]\|^\[lockstep\] batch" gpurun_out/sweep_ls_err.txt | cut -c1-120; }
for rep in 1 2; do
run IPM_LS_DEBUG=1
run IPM_LS_DEBUG=1 GPU_MAX_HW_QUEUES=8
run IPM_LS_DEBUG=1 GPU_MAX_HW_QUEUES=8 IPM_LOCKSTEP_CLASSIC_THREADS=2
run IPM_LS_DEBUG=1 GPU_MAX_HW_QUEUES=8 IPM_LOCKSTEP_CLASSES=1000,2200,3500 IPM_LOCKSTEP_CLASSIC_THREADS=2
run IPM_LS_DEBUG=1 GPU_MAX_HW_QUEUES=6 IPM_LOCKSTEP_CLASSES=2200,3500,5000
run IPM_LS_DEBUG=1 IPM_LOCKSTEP_CLASSIC_THREADS=2
run IPM_LS_DEBUG=1 IPM_SP_MODE=task
run IPM_LOCKSTEP=0
run IPM_LOCKSTEP=0 GPU_MAX_HW_QUEUES=8
done
