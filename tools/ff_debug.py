"""Fused formation + factorization against the serial path at one size: where do the factors differ?  (GPU box)
   python tools/ff_debug.py m n [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import interiorpointmethod_amd as ipm                              # noqa: E402
from interiorpointmethod_amd.workloads import synthetic_lp         # noqa: E402

m, n = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
A, b, c = synthetic_lp(m, n, seed=5)


def run(mode):
    os.environ["IPM_FUSED_FACTOR"] = mode
    with ipm.IpmSolver(A, b, c) as sv:
        sv.init_state(0.0)
        st = sv.iterate(steps)
        L = sv.get_factor()
        sch = sv.schedule()
    return st, L, sch


st0, L0, s0 = run("0")
st1, L1, s1 = run("force")
print("serial:", s0, "\nfused :", s1)
nb = (m + 127) // 128
E = np.zeros((nb, nb))
for i in range(nb):
    for j in range(i + 1):
        a, b_ = L0[i * 128:(i + 1) * 128, j * 128:(j + 1) * 128], L1[i * 128:(i + 1) * 128, j * 128:(j + 1) * 128]
        E[i, j] = np.max(np.abs(a - b_)) / max(1e-300, np.max(np.abs(L0)))
np.set_printoptions(linewidth=250, precision=1)
print("relative difference per tile (rows = block row):")
print(E)
print("objective", st0["objective"], st1["objective"], "pivots fixed", st0["pivots_fixed"], st1["pivots_fixed"])
