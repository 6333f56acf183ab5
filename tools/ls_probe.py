"""One lockstep batch of the named LPs, timed: python tools/ls_probe.py NAME ...  (GPU box; IPM_LS_PROF=1 / IPM_LS_DEBUG=1 for detail)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import interiorpointmethod_amd as ipm
from interiorpointmethod_amd.matio import load_npz_problem
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "netlib")
names = sys.argv[1:]
svs = []
for nm in names:
    A, b, c, _, _ = load_npz_problem(os.path.join(G, nm + ".npz"))
    sv = ipm.IpmSolver(A, b, c, lockstep=True, concurrent=True); sv.init_state(1.0); svs.append(sv)
t0 = time.perf_counter()
st = ipm.solve_lockstep(svs, tol=1e-8, max_iter=300)
dt = time.perf_counter() - t0
print("batch of %d: %.3f s" % (len(names), dt))
for nm, s_, sv in zip(names, st, svs):
    print("  %-9s blocks %2d it %3d status %d" % (nm, sv.schedule()["blocks"], s_["iterations"], s_["status"]))
    sv.close()
