# potrf_diag: phase stamps (tools/potrf_stamps.py, twice; once without the write-back stores), the GPU tests, the 73-LP suite and the dense line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do python tools/potrf_stamps.py 2>&1 | tail -4; done > gpurun_out/r04_potrf_stamps_tail.txt; cat gpurun_out/r04_potrf_stamps_tail.txt
IPM_POTRF_SKIP=3 python tools/potrf_stamps.py 2>&1 | tail -3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_potrf_tail_pytest.log 2>&1; tail -3 gpurun_out/r04_potrf_tail_pytest.log
for i in 1 2 3; do python bench.py --workload netlib --no-cpu-baseline --netlib-set all 2>/dev/null | tail -1 | cut -c1-120; done
python bench.py --no-netlib --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-700
python tools/ls_probe.py PILOT87 | head -3
