#!/bin/bash
# multi-step block substitutions: bitwise test, then the batched suite with and without
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "multi_step or cholesky_and_solve or config4 or netlib_parity or two_handles or drop_in" > $O/m_pytest.log 2>&1 || { tail -30 $O/m_pytest.log; exit 1; }
tail -1 $O/m_pytest.log
for T in 0 1 0 1; do
  IPM_TRSV_MULTI=$T timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/m_netlib_$T.json 2> $O/m_netlib_$T.err || { tail -5 $O/m_netlib_$T.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/m_netlib_$T.json').read().strip().splitlines()[-1]); s=d['summary']
print('IPM_TRSV_MULTI=$T: %.2f LPs/s wall %.3f converged %d iterations %d' % (d['value'], d['wall_seconds'], s['converged'], s['total_iterations']))"
done
for T in 0 2 4; do
  IPM_SS_TINY_TILES=$T timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/m_tiny_$T.json 2> $O/m_tiny_$T.err || { tail -5 $O/m_tiny_$T.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/m_tiny_$T.json').read().strip().splitlines()[-1]); s=d['summary']
print('IPM_SS_TINY_TILES=$T: %.2f LPs/s wall %.3f converged %d iterations %d' % (d['value'], d['wall_seconds'], s['converged'], s['total_iterations']))"
done
