#!/bin/bash
# multi-step block substitutions with double-buffered block streaming: bitwise test, lone LPs, the batched suite
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "multi_step or cholesky_and_solve" > $O/p_pytest.log 2>&1 || { tail -30 $O/p_pytest.log; exit 1; }
tail -1 $O/p_pytest.log
for T in 0 1; do
  for NM in BNL2 FINNIS DEGEN3; do IPM_TRSV_MULTI=$T python3 tools/ss_timeline.py $NM 40 2>&1 | tail -1 | sed "s/^/IPM_TRSV_MULTI=$T single-stream /"; done
done
for T in 0 1 0 1; do
  IPM_TRSV_MULTI=$T timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/p_netlib_$T.json 2> $O/p_netlib_$T.err || { tail -5 $O/p_netlib_$T.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/p_netlib_$T.json').read().strip().splitlines()[-1]); s=d['summary']
print('IPM_TRSV_MULTI=$T: %.2f LPs/s wall %.3f converged %d iterations %d' % (d['value'], d['wall_seconds'], s['converged'], s['total_iterations']))"
done
