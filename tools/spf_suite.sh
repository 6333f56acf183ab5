#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for g in 64 256 100000; do
  IPM_SP_GRID=$g timeout -k 10 400 python bench.py --workload netlib --netlib-set all --no-cpu-baseline > gpurun_out/all_g$g.json 2> gpurun_out/all_g$g.err || { tail -5 gpurun_out/all_g$g.err; exit 1; }
  IPM_SP_GRID=$g timeout -k 10 400 python bench.py --workload netlib --netlib-set general --start-point mehrotra --no-cpu-baseline > gpurun_out/genm_g$g.json 2> gpurun_out/genm_g$g.err || { tail -5 gpurun_out/genm_g$g.err; exit 1; }
done
python - <<'PY'
import json
for g in (64,256,100000):
  for f in ("all","genm"):
    d=json.loads(open("gpurun_out/%s_g%d.json"%(f,g)).read().strip().splitlines()[-1])
    p=d["per_lp"]
    print(f, g, "value=%.3f LPs/s wall=%.3f conv=%d its=%d"%(d["value"], d["wall_seconds"], d["summary"]["converged"], d["summary"]["total_iterations"]), {k:p[k]["s"] for k in ("STOCFOR3","STOCFOR2","CZPROB","SCTAP3","SIERRA","80BAU3B","SHELL")})
PY
