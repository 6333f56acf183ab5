#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --workload netlib --no-cpu-baseline --netlib-set $SET 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %.2f LPs/s wall %.3f s converged %d slowest %s' % (d['value'], d['wall_seconds'], d['summary']['converged'], d.get('slowest_lp')))"; }
for rep in 1 2; do
for k in 0 1 2 3; do SET=all run IPM_STREAM_SKIP=$k; done
for k in 0 1 2 3; do SET=parity run IPM_STREAM_SKIP=$k; done
done
