#!/usr/bin/env python3
"""One mid-size LP on a SINGLE-STREAM handle (what the batched mode creates), a few iterations -- run under
rocprofv3 --kernel-trace to see the dependent chain of one iteration:   python3 tools/ss_timeline.py [NAME] [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interiorpointmethod_amd as ipm
from interiorpointmethod_amd.matio import load_npz_problem
name = sys.argv[1] if len(sys.argv) > 1 else "DEGEN3"
its = int(sys.argv[2]) if len(sys.argv) > 2 else 24
A, b, c, cTlb, valid = load_npz_problem(os.path.join(ROOT, "tests", "golden", "netlib", name + ".npz"))
with ipm.IpmSolver(A, b, c, concurrent=True) as sv:
    sv.init_state(1.0)
    st = sv.solve(tol=1e-8, max_iter=its)
    print(name, st["status"], st["iterations"], "%.3f ms per iteration" % (st["solve_ms"] / max(1, st["iterations"])))
