"""Randomised size sweep: HIP solve vs the NumPy oracle (objective, iteration count) on dense and sparse LPs whose
sizes straddle the 128-row block and 1024-row group boundaries."""
import os, sys, time
import numpy as np
from scipy import sparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import interiorpointmethod_amd as ipm
from oracle import ipm_oracle as O

rng = np.random.default_rng(2026)
bad = 0
sizes = [(1, 3), (2, 2), (127, 300), (128, 256), (129, 257), (255, 700), (257, 513), (640, 900), (1000, 2100), (1023, 2047),
         (1024, 2048), (1025, 2051), (1500, 1600), (2047, 4100), (2049, 4097), (2304, 4700), (3000, 6100),
         (6200, 9000), (7000, 7100)]      # >= 48 blocks: the two-level factorization; sparse: RCM + tile envelope
for (m, n) in sizes:
    for kind in ("dense", "sparse"):
        if kind == "dense":
            A = rng.standard_normal((m, n))
        else:
            A = sparse.random(m, n, density=min(1.0, 6.0 / n + 0.002), random_state=np.random.RandomState(m + n), format="csr")
            A = sparse.csc_matrix(A + sparse.eye(m, n, format="csr"))
        x0 = rng.uniform(0.5, 1.5, (n, 1)); y0 = rng.standard_normal((m, 1)); s0 = rng.uniform(0.5, 1.5, (n, 1))
        b = A @ x0; c = A.T @ y0 + s0
        t = time.time()
        x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, y0=0.0, max_iter=300)
        tg = time.time() - t
        if m <= 1100:
            Ad = A.toarray() if sparse.issparse(A) else A
            xo, yo, so, io = O.solve(Ad, b, c, tol=1e-8, y0=0.0, method="normal", max_iter=300)
            ok = info["status"] == 1 and abs(info["objective"] - io["objective"]) <= 1e-6 * max(1, abs(io["objective"])) and abs(info["iterations"] - io["iterations"]) <= 2
            ref = "oracle it=%d" % io["iterations"]
        else:   # size-independent properties only
            Ad = A
            rb = np.linalg.norm(A @ x - b) / (1 + np.linalg.norm(b)); rc = np.linalg.norm(A.T @ y + s - c) / (1 + np.linalg.norm(c))
            ok = info["status"] == 1 and rb < 1e-7 and rc < 1e-7 and float((x * s).sum()) < 1e-7 and abs(float((c * x).sum()) - float((b * y).sum())) < 1e-6 * (1 + abs(info["objective"]))
            ref = "rb %.1e rc %.1e" % (rb, rc)
        bad += (not ok)
        print("%-6s %5d x %5d  it=%3d st=%d obj=% .10e  %s  %.2fs %s" % (kind, m, n, info["iterations"], info["status"], info["objective"], ref, tg, "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
