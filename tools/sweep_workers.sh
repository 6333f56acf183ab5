#!/bin/bash
# usage: tools/sweep_workers.sh [set] w1 w2 ...  -> Netlib suite wall time and LPs/s per number of LPs in flight on one GPU
SET=${1:-all}; shift
for W in "$@"; do
  timeout -k 10 300 python bench.py --workload netlib --netlib-set $SET --workers $W --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('workers $W: %.3f LPs/s, %.2f s, %d converged, %d iterations' % (d['value'], d['wall_seconds'], d['summary']['converged'], d['summary']['total_iterations']))"
done
