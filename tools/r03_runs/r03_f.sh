#!/bin/bash
# fused path: formation chunks per ordinary tile (IPM_FF_Q) around the default of 4
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
run() {
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-netlib --no-cpu-baseline --steps 40 > $O/f2_$name.json 2> $O/f2_$name.err || { tail -3 $O/f2_$name.err; return 1; }
  python - $name <<'PY'
import json,sys
d=json.loads(open("gpurun_out/f2_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("%-20s it/s %7.2f ms %.3f %s worker-kernel ms %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["objective_check"], d["phases_ms_per_step"]["form"]))
PY
}
run q4
run q2 IPM_FF_Q=2
run q3 IPM_FF_Q=3
run q5 IPM_FF_Q=5
run q4_batch6 IPM_FF_BATCH=6
run q4_batch3 IPM_FF_BATCH=3
run q4_window3 IPM_FF_WINDOW=3
