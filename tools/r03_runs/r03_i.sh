#!/bin/bash
# full GPU suite + default bench line after the sparse-factor changes
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/i_pytest.log 2>&1 || { tail -40 $O/i_pytest.log; exit 1; }
tail -2 $O/i_pytest.log
timeout -k 10 600 python bench.py > $O/i_bench.json 2> $O/i_bench.err || { tail -20 $O/i_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/i_bench.json").read().strip().splitlines()[-1])
print("dense", round(d["value"], 2), "it/s; roofline", round(d["roofline"]["frac"], 4), d["roofline"]["traffic"], d["objective_check"])
for k in ("netlib_all", "netlib"):
    n = d[k]
    print(k, round(n["value"], 2), "LPs/s wall", round(n["wall_seconds"], 3), {q: n["summary"][q] for q in ("n", "converged", "timeouts_recovered", "serial_launches", "setup_seconds_sum")}, n["roofline"].get("sparse_factor_lps"))
PY
