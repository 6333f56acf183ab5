#!/bin/bash
# batched 73-LP suite (8 LPs in flight): fewer dependent launches per iteration through the single-launch substitutions
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
run() {
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/g2_$name.json 2> $O/g2_$name.err || { tail -5 $O/g2_$name.err; return 1; }
  python - $name <<'PY'
import json,sys
d=json.loads(open("gpurun_out/g2_%s.json"%sys.argv[1]).read().strip().splitlines()[-1]); s=d["summary"]
print("%-28s %.2f LPs/s wall %.3f s converged %d iterations %d recovered %d" % (sys.argv[1], d["value"], d["wall_seconds"], s["converged"], s["total_iterations"], s["timeouts_recovered"]))
PY
}
run default
run persistent_where_ungrouped IPM_PERSISTENT_TRSV=1
run persistent_everywhere IPM_PERSISTENT_TRSV=1 IPM_GROUPED_TRSV=0
run default_again
