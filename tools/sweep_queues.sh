#!/bin/bash
# usage: tools/sweep_queues.sh [set]  -> Netlib suite LPs/s for GPU_MAX_HW_QUEUES x workers (hardware queues the HIP runtime may use)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
SET=${1:-all}
for Q in 4 8 16; do
  for W in 8 12; do
    GPU_MAX_HW_QUEUES=$Q timeout -k 10 300 python bench.py --workload netlib --netlib-set $SET --workers $W --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $Q workers $W: %.3f LPs/s, %.2f s, %d converged, %d iterations' % (d['value'], d['wall_seconds'], d['summary']['converged'], d['summary']['total_iterations']))"
  done
done
