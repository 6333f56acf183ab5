#!/bin/bash
# usage: tools/netlib_errs.sh W N -> N runs of the 73-LP suite with W LPs in flight: summary + distinct stderr lines of failed LPs
for i in $(seq 1 $2); do
  timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers $1 --no-cpu-baseline 2> /tmp/netlib_err_$i.log | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['summary']; print('run $i: %.2f s converged %d errors %d nan %d cap %d iterations %d' % (d['wall_seconds'], s['converged'], s['errors'], s['nan'], s['max_iter'], s['total_iterations']))"
  grep "batch\]" /tmp/netlib_err_$i.log | sed 's/[0-9]* x [0-9]* LP failed/LP failed/' | sort | uniq -c | sort -rn | head -5
done
