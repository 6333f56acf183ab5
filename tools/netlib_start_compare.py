"""Reference start (x = s = 1) vs Mehrotra's starting point over the Netlib fixtures: status, iterations, objective
error against the Netlib optimum table."""
import glob, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interiorpointmethod_amd import solver as S
from interiorpointmethod_amd.matio import load_npz_problem
opt = json.load(open(os.path.join("tests", "golden", "netlib_optima.json")))
tot = {"reference": [0, 0, 0.0], "mehrotra": [0, 0, 0.0]}
for f in sorted(glob.glob(os.path.join("tests", "golden", "netlib", "*.npz"))):
    nm = os.path.basename(f)[:-4]
    A, b, c, cTlb, valid = load_npz_problem(f)
    if not valid:
        continue
    row = "%-10s" % nm
    for start in ("reference", "mehrotra"):
        t = time.perf_counter()
        try:
            x, y, s, info = S.solve_with_info(A, b, c, tol=1e-8, max_iter=300, y0=1.0, start=start)
        except Exception as e:
            row += "  %s: ERROR %s" % (start, str(e)[:40]); continue
        dt = time.perf_counter() - t
        obj = info["objective"] - cTlb
        o = opt.get(nm)
        err = abs(obj - o) / max(1.0, abs(o)) if (o is not None and np.isfinite(obj)) else float("nan")
        good = info["status"] == 1 and (o is None or err <= 1e-6)
        tot[start][0] += info["status"] == 1; tot[start][1] += good; tot[start][2] += dt
        row += "  %s: %-9s it=%3d err=%8.1e %5.2fs" % (start[:3], info["status_name"], info["iterations"], err, dt)
    print(row, flush=True)
print("converged / converged-to-optimum / seconds:", tot)
