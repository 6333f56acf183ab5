#!/bin/bash
# 73-LP Netlib suite on one GPU, dense-tile factor only vs the factor="auto" rule (sparse multifrontal factor where it pays)
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for f in dense auto; do
  IPM_FACTOR=$f timeout -k 10 300 python bench.py --workload netlib --netlib-set all --no-cpu-baseline > gpurun_out/suite_$f.json 2> gpurun_out/suite_$f.err || { tail -5 gpurun_out/suite_$f.err; exit 1; }
done
python - <<'PY'
import json
for f in ("suite_dense", "suite_auto"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, "value=%.3f LPs/s wall=%.3f" % (d["value"], d["wall_seconds"]), d["summary"])
PY
