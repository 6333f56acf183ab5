#!/bin/bash
# 73-LP suite under the lockstep batches, a few settings each (GPU box): tools/sweep_ls3.sh > gpurun_out/sweep_ls3.txt
run() { echo "== $*"; env "$@" timeout -k 10 150 python bench.py --workload netlib --no-cpu-baseline $SET 2>gpurun_out/sweep_ls_err.txt | tee gpurun_out/sweep_last.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  %.2f LPs/s wall %.3f s converged %d slowest %s it %d timeouts %d' % (d['value'], d['wall_seconds'], d['summary']['converged'], d.get('slowest_lp'), d['summary']['total_iterations'], d['summary']['timeouts_recovered']))"; grep "^\[batch\]\|^\[lockstep\] batch" gpurun_out/sweep_ls_err.txt | cut -c1-150
python - <<'PY'
import json
for l in open('gpurun_out/sweep_last.log'):
    if l.startswith('BENCH_DETAIL'):
        p = json.loads(l[len('BENCH_DETAIL'):].strip())['per_lp']
        t = sorted(((v['s'], k, v['it'], v['setup_s'], v['solve_s']) for k, v in p.items()), reverse=True)[:5]
        print('   ', '; '.join('%s %.3f (setup %.3f, %d it)' % (k, s, su, it) for s, k, it, su, so in t))
PY
}
for rep in 1 2; do
SET="--netlib-set all" run IPM_LS_DEBUG=1
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_SP_MODE=task
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_SP_MODE=task IPM_LOCKSTEP_DENSE_ROWS=4000
SET="--netlib-set all" run IPM_LS_DEBUG=1 IPM_LOCKSTEP_DENSE_ROWS=4000
done
