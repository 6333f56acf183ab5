import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import interiorpointmethod_amd as ipm
from interiorpointmethod_amd.matio import load_npz_problem
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "netlib")
for name, ref in (("QAP8", 203.5), ("QAP12", 522.89435056), ("QAP15", 1040.9940410)):
    A, b, c, cTlb, valid = load_npz_problem(os.path.join(G, name + ".npz"))
    for reg in (0.0, 1e-14, 1e-12, 1e-10):
        t0 = time.time()
        x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, max_iter=120, regularize=reg)
        print("%-6s reg=%g it=%d st=%s obj=%.10f rel=%.2e rp=%.1e rd=%.1e gap=%.1e fixed=%d %.2fs" % (
            name, reg, info["iterations"], info["status_name"], info["objective"], abs(info["objective"] - ref) / ref,
            info["rp"], info["rd"], info["gap"], info["pivots_fixed"], time.time() - t0), flush=True)
