#!/bin/bash
# hardware queues x LPs in flight for the batched Netlib suite: tools/sweep_queues.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for q in 4 8 16; do for w in 8 12 16; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --workload netlib --no-cpu-baseline --workers $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('queues $q workers $w: %.2f LPs/s  wall %.3f s  converged %s  slowest %.3f (%s)' % (d['value'], d['wall_seconds'], d['summary']['converged'], d['projected_makespan_8gpu_s'], d['slowest_lp']))"
done; done
