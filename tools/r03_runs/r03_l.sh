#!/bin/bash
# single-stream handles (batched suite): narrow-tile panel / update kernels for the last trailing blocks
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cholesky_and_solve or config4 or netlib_parity or two_handles" > $O/l_pytest0.log 2>&1 || { tail -30 $O/l_pytest0.log; exit 1; }
tail -1 $O/l_pytest0.log
IPM_SS_SMALL_TILES=64 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "cholesky_and_solve or config4 or netlib_parity or two_handles" > $O/l_pytest1.log 2>&1 || { tail -30 $O/l_pytest1.log; exit 1; }
tail -1 $O/l_pytest1.log
for T in 0 4 8 16 64 0; do
  IPM_SS_SMALL_TILES=$T timeout -k 10 300 python bench.py --workload netlib --netlib-set all --workers 8 --no-cpu-baseline > $O/l_netlib_$T.json 2> $O/l_netlib_$T.err || { tail -5 $O/l_netlib_$T.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/l_netlib_$T.json').read().strip().splitlines()[-1]); s=d['summary']
print('IPM_SS_SMALL_TILES=$T: %.2f LPs/s wall %.3f converged %d iterations %d' % (d['value'], d['wall_seconds'], s['converged'], s['total_iterations']))"
done
