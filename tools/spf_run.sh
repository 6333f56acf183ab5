#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export IPM_SP_XCD_REPORT=1
for x in 0 1; do
IPM_SP_XCD=$x IPM_SP_XCD_PROTO=0 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
from interiorpointmethod_amd.matio import load_npz_problem
from interiorpointmethod_amd.solver import IpmSolver
A,b,c,_,v=load_npz_problem("tests/golden/netlib/STOCFOR2.npz")
for k in range(3):
    with IpmSolver(sp.csc_matrix(A),b,c,factor="sparse") as sv:
        z=sv.normal_solve(np.ones(A.shape[0]))
        fi=sv.factor_info()
        print("IPM_SP_XCD=%s handle %d: xcc mask = %s grid-ish tasks=%d" % (os.environ["IPM_SP_XCD"], k, bin(fi["serial_launches"]>>32), fi["tasks"]))
PY
done
