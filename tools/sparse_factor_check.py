#!/usr/bin/env python3
"""Sparse multifrontal Cholesky vs the dense-tile path on Netlib fixtures: factor / solve parity against the host, end-to-end
objective and time per iteration both ways.   python tools/sparse_factor_check.py [--kernel] NAME ..."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interiorpointmethod_amd as ipm                                    # noqa: E402
from interiorpointmethod_amd.matio import load_npz_problem               # noqa: E402
from interiorpointmethod_amd.solver import IpmSolver                     # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def kernel_level(nm, A, b, c):
    rng = np.random.default_rng(1)
    d = rng.uniform(0.5, 2.0, A.shape[1])
    rhs = rng.standard_normal(A.shape[0])
    B = (A @ sp.diags(d) @ A.T).toarray()
    with IpmSolver(A, b, c, factor="sparse") as sv:
        fi = sv.factor_info()
        z = sv.normal_solve(rhs, d=d)
        fixed = sv.last_pivots_fixed
        L = sv.get_factor()
        z2 = sv.normal_solve(rhs, d=d)
        perm = sv._perm
    Bp = B[np.ix_(perm, perm)]
    lerr = np.abs(L @ L.T - Bp).max() / np.abs(Bp).max()
    res = np.linalg.norm(B @ z.ravel() - rhs) / np.linalg.norm(rhs)
    print("%-10s kernel: %s fixed=%d |LL^T-B|/|B|=%.2e  |Bz-r|/|r|=%.2e  repeat_bitwise=%s" %
          (nm, fi, fixed, lerr, res, np.array_equal(z, z2)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--kernel", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--no-sparse", action="store_true")
    args = ap.parse_args()
    for nm in args.names:
        A, b, c, cTlb, valid = load_npz_problem(os.path.join(G, "netlib", nm + ".npz"))
        A = sp.csc_matrix(A)
        if args.kernel:
            kernel_level(nm, A, b, c)
        e2e = os.path.join(G, "e2e_%s.npz" % nm)
        ref = float(np.load(e2e)["objective"]) if os.path.exists(e2e) else float("nan")
        for factor in (("sparse",) if args.no_dense else (("dense",) if args.no_sparse else ("sparse", "dense"))):
            t0 = time.time()
            x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, max_iter=300, factor=factor)
            wall = time.time() - t0
            print("%-10s %-6s status=%s it=%d obj=%.12e ref=%.12e rel=%.1e fixed=%d  solve_ms=%.1f (%.3f ms/it) wall=%.2fs" %
                  (nm, factor, info["status_name"], info["iterations"], info["objective"], ref,
                   abs(info["objective"] - ref) / max(1.0, abs(ref)), info["pivots_fixed"], info["solve_ms"],
                   info["solve_ms"] / max(1, info["iterations"]), wall), flush=True)


if __name__ == "__main__":
    main()
