#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> one bench line (it/s, phases) per value of the environment variable
VAR=$1; shift
for V in "$@"; do
  r=$(env $VAR=$V timeout -k 5 120 python bench.py --no-cpu-baseline --no-netlib --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; print('%.1f it/s form %.3f factor %.3f tri %.3f other %.3f device_total %.3f %s' % (d['value'], p['form'], p['factor'], p['trisolve'], p['other'], p['device_total'], d.get('objective_check')))")
  echo "$VAR=$V : $r"
done
