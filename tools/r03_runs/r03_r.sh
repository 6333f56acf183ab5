#!/bin/bash
# potrf_diag with interleaved panel-row substitutions: tests, phase stamps, bench
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_factor.py -x -q > $O/r_pytest.log 2>&1 || { tail -30 $O/r_pytest.log; exit 1; }
tail -1 $O/r_pytest.log
timeout -k 10 200 python tools/potrf_stamps.py > $O/r_potrf_stamps.log 2>&1; tail -14 $O/r_potrf_stamps.log | cut -c1-200
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > $O/r_bench.json 2> $O/r_bench.err || { tail -5 $O/r_bench.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/r_bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],2), d['ms_per_step'], d['objective_check'], d['phases_ms_per_step']['form'], round(d['netlib_all']['value'],2), round(d['netlib']['value'],2))"
for NM in BNL2 FINNIS DEGEN3; do python3 tools/ss_timeline.py $NM 40 2>&1 | tail -1; done
