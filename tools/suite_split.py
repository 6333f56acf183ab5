#!/usr/bin/env python3
"""Where does the wall time of the Netlib suite go on one GPU, one LP at a time: host setup (CSC clean-up, row ordering,
handle creation, upload, product list) vs the device solve (ipm_stats.solve_ms)."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interiorpointmethod_amd as ipm
from interiorpointmethod_amd.matio import load_npz_problem
rows = []
ipm.solve_with_info(*load_npz_problem(os.path.join(ROOT, "tests", "golden", "netlib", "AFIRO.npz"))[:3])
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "netlib", "*.npz"))):
    A, b, c, cTlb, valid = load_npz_problem(f)
    if not valid:
        continue
    t0 = time.perf_counter()
    x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, max_iter=300)
    rows.append((os.path.basename(f)[:-4], A.shape[0], time.perf_counter() - t0, info["solve_ms"] * 1e-3, info["iterations"]))
tw = sum(r[2] for r in rows); td = sum(r[3] for r in rows)
print("73 LPs: wall %.2f s, device solve %.2f s, host setup/teardown %.2f s (%.0f %%)" % (tw, td, tw - td, 100 * (tw - td) / tw))
for r in sorted(rows, key=lambda r: -(r[2] - r[3]))[:12]:
    print("  %-10s m=%5d wall %.3f device %.3f host %.3f" % (r[0], r[1], r[2], r[3], r[2] - r[3]))
