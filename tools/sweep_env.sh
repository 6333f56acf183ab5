#!/bin/bash
# one bench line per value of an environment variable: tools/sweep_env.sh VAR v1 v2 ... [-- bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${VALS[@]}"; do
  env $VAR=$v timeout -k 10 300 python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v: %.2f %s  wall %.3f s  converged %s  makespan %.3f (%s)' % (d['value'], d['unit'], d.get('wall_seconds', 0), d.get('summary', {}).get('converged'), d.get('projected_makespan_8gpu_s', 0), d.get('slowest_lp')))"
done
