#!/bin/bash
# the batched suite against the number of hardware queues the HIP runtime uses (GPU_MAX_HW_QUEUES, default 4), 8 and 12 LPs in flight
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
for Q in 4 6 8; do
  for W in 8 12; do
    GPU_MAX_HW_QUEUES=$Q timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers $W --no-cpu-baseline > $O/u_netlib_q${Q}_w$W.json 2> $O/u_netlib_q${Q}_w$W.err || { tail -5 $O/u_netlib_q${Q}_w$W.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/u_netlib_q${Q}_w$W.json').read().strip().splitlines()[-1]); s=d['summary']
print('GPU_MAX_HW_QUEUES=$Q workers $W: %.2f LPs/s wall %.3f converged %d' % (d['value'], d['wall_seconds'], s['converged']))"
  done
done
