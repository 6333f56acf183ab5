#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> one bench line (it/s, phases) per value of the environment variable
VAR=$1; shift
for V in "$@"; do
  r=$(env $VAR=$V timeout -k 5 120 python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phases_ms_per_step']; print('%.1f it/s form %.2f factor %.2f tri %.2f other %.2f' % (d['value'], p['form'], p['factor'], p['trisolve'], p['other']))")
  echo "$VAR=$V : $r"
done
