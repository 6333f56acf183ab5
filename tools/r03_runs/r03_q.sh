#!/bin/bash
# soak of the fused path: 4000 iterations at the headline size, then the recovered-time-out counter of the handle
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python - <<'PY'
import time, numpy as np
import interiorpointmethod_amd as ipm
from interiorpointmethod_amd.workloads import synthetic_lp
A, b, c = synthetic_lp(4096, 8192, seed=0)
with ipm.IpmSolver(A, b, c) as sv:
    tot = 0; t0 = time.time()
    for rep in range(200):
        sv.init_state(0.0)
        st = sv.solve(tol=1e-30, max_iter=20)        # 20 iterations from the start point, never converges at this tolerance
        tot += st["iterations"]
        if rep % 50 == 0: print(rep, st["iterations"], st["objective"], sv.schedule(), flush=True)
    dt = time.time() - t0
    sch = sv.schedule()
    print("iterations", tot, "wall %.2f s" % dt, "it/s %.1f" % (tot / dt), "schedule", sch)
    assert sch["timeouts_recovered"] == 0 and sch["fused_factor"] == 1
PY
