#!/bin/bash
# potrf_diag factors only the panels of a diagonal block that hold rows of the LP: tests, lone LPs, suite
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/z_pytest.log 2>&1 || { tail -30 $O/z_pytest.log; exit 1; }
tail -1 $O/z_pytest.log
for NM in DEGEN3 FINNIS 25FV47 BNL1 BNL2; do python3 tools/ss_timeline.py $NM 40 2>&1 | tail -1; done
for T in 1 2 3; do
  timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('suite run $T: %.2f LPs/s wall %.3f iterations %d converged %d' % (d['value'], d['wall_seconds'], d['summary']['total_iterations'], d['summary']['converged']))"
done
