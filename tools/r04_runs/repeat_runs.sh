# the default bench line's Netlib legs run three times in one process: do repeated runs keep their speed?  (stream reuse, batch.py)
cd "$GRAFT_REPO_ROOT"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('dense %.1f it/s' % d['value'])
for k in ('netlib_all', 'netlib'):
    print(k, d[k]['value'], d[k]['wall_seconds_runs'], d[k]['slowest_lp'])"
timeout -k 10 600 python -m pytest tests/test_gpu_lockstep.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
