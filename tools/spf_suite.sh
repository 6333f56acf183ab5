#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
IPM_FACTOR=dense timeout -k 10 300 python bench.py --workload netlib --netlib-set all --no-cpu-baseline > gpurun_out/suite_dense.json 2> gpurun_out/suite_dense.err || { tail -5 gpurun_out/suite_dense.err; exit 1; }
timeout -k 10 300 python bench.py --workload netlib --netlib-set all --no-cpu-baseline > gpurun_out/suite_auto.json 2> gpurun_out/suite_auto.err || { tail -5 gpurun_out/suite_auto.err; exit 1; }
timeout -k 10 300 python bench.py --workload netlib --netlib-set all --no-cpu-baseline > gpurun_out/suite_auto2.json 2> gpurun_out/suite_auto2.err || { tail -5 gpurun_out/suite_auto2.err; exit 1; }
python - <<'PY'
import json
for f in ("suite_dense","suite_auto","suite_auto2"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, "value=%.3f LPs/s wall=%.3f"%(d["value"], d["wall_seconds"]), d["summary"])
PY
