#!/bin/bash
# the GPU suite under the non-default switches a recovered time-out or a shared device selects by itself
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out; mkdir -p $O
IPM_FUSED_FACTOR=0 timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not fused" > $O/s_pytest_serial.log 2>&1 || { tail -30 $O/s_pytest_serial.log; exit 1; }
echo "IPM_FUSED_FACTOR=0: $(tail -1 $O/s_pytest_serial.log)"
IPM_SP_MODE=level timeout -k 10 900 python -m pytest tests/test_gpu_sparse_factor.py tests/test_gpu_parity.py -x -q -k "not schedule_independence and not forward_substitution" > $O/s_pytest_level.log 2>&1 || { tail -30 $O/s_pytest_level.log; exit 1; }
echo "IPM_SP_MODE=level: $(tail -1 $O/s_pytest_level.log)"
IPM_FLAG_SYNC=0 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/s_pytest_events.log 2>&1 || { tail -30 $O/s_pytest_events.log; exit 1; }
echo "IPM_FLAG_SYNC=0 (stream events): $(tail -1 $O/s_pytest_events.log)"
