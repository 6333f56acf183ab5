#!/bin/bash
# fused path: tests, then bench lines at the headline size for the shipped configuration and its variants
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out; mkdir -p $O
IPM_FF_DEBUG=1 timeout -k 10 600 python -m pytest tests/test_gpu_fused_factor.py -x -q > $O/c_pytest.log 2>&1 || { tail -60 $O/c_pytest.log; exit 1; }
tail -1 $O/c_pytest.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-netlib --no-cpu-baseline --steps 40 > $O/c_$name.json 2> $O/c_$name.err || { tail -3 $O/c_$name.err; return 1; }
  python - $name <<'PY'
import json,sys
d=json.loads(open("gpurun_out/c_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s it/s %7.2f ms %.3f %s worker-kernel ms %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["objective_check"], d["phases_ms_per_step"]["form"]))
PY
}
run serial IPM_FUSED_FACTOR=0
run ff_default
run ff_first4 IPM_FF_Q_FIRST=4 IPM_FF_Q_SECOND=4
run ff_first16_16 IPM_FF_Q_FIRST=16 IPM_FF_Q_SECOND=16
run ff_first8_4 IPM_FF_Q_FIRST=8 IPM_FF_Q_SECOND=4
run ff_colmajor IPM_FF_ROW_WEIGHT=0 IPM_FF_COL_WEIGHT=1
run ff_w2140 IPM_FF_ROW_WEIGHT=21 IPM_FF_COL_WEIGHT=40
run ff_nostagger IPM_FF_STAGGER=0
run ff_q6 IPM_FF_Q=6
