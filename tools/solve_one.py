#!/usr/bin/env python3
"""Solve named Netlib fixtures one after another on one GPU (for rocprofv3 --kernel-trace --stats):
    python tools/solve_one.py DEGEN3 SHELL BNL2 [--max-iter 300]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interiorpointmethod_amd as ipm                      # noqa: E402
from interiorpointmethod_amd.matio import load_npz_problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--max-iter", type=int, default=300)
args = ap.parse_args()
for nm in args.names:
    A, b, c, cTlb, valid = load_npz_problem(os.path.join(ROOT, "tests", "golden", "netlib", nm + ".npz"))
    t0 = time.time()
    x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, max_iter=args.max_iter)
    print("%-10s m=%d n=%d it=%d status=%s obj=%.10e  %.3f s  (%.3f ms/it device)" % (
        nm, A.shape[0], A.shape[1], info["iterations"], info["status_name"], info["objective"], time.time() - t0,
        info["solve_ms"] / max(info["iterations"], 1)))
