#!/usr/bin/env python3
"""Results table in the layout of the reference's script.py:139-198 (conclusion1.txt), GPU column instead
of the reference's own solver:  Name | Interi(GPU) time | Obj fun (reference, golden) | Interi(GPU) obj | its | status

    python tools/netlib_report.py [--out conclusion_gpu.txt] [--regularize 0] [NAME ...]

Inputs are the committed Netlib fixtures (tests/golden/netlib/*.npz); the reference objective column comes
from the golden end-to-end vectors where the reference converges (tests/golden/e2e_*.npz), else blank.
"""
import argparse
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interiorpointmethod_amd as ipm                      # noqa: E402
from interiorpointmethod_amd.matio import load_npz_problem  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--out", default="conclusion_gpu.txt")
    ap.add_argument("--regularize", type=float, default=0.0)
    ap.add_argument("--max-iter", type=int, default=300)
    args = ap.parse_args()
    names = args.names or sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(G, "netlib", "*.npz")))
    with open(args.out, "w") as fh:
        fh.write("{0:17s} {1:>14s} {2:>22s} {3:>22s} {4:>6s} {5:>10s}\r\n".format(
            "Name", "Interi time", "Obj fun (reference)", "Interi (MI355X)", "its", "status"))
        for nm in names:
            A, b, c, cTlb, valid = load_npz_problem(os.path.join(G, "netlib", nm + ".npz"))
            ref = ""
            e2e = os.path.join(G, "e2e_%s.npz" % nm)
            if os.path.exists(e2e):
                ref = "%.10e" % float(np.load(e2e)["objective"])
            if not valid:
                fh.write("{0:17s} {1:>14s} {2:>22s} {3:>22s} {4:>6s} {5:>10s}\r\n".format(nm, "-", ref, "-", "-", "invalid"))
                continue
            t0 = time.time()
            x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, max_iter=args.max_iter, regularize=args.regularize)
            fh.write("{0:17s} {1:14.3f} {2:>22s} {3:22.10e} {4:6d} {5:>10s}\r\n".format(
                nm, time.time() - t0, ref, info["objective"], info["iterations"], info["status_name"]))
            fh.flush()
    print(open(args.out).read())


if __name__ == "__main__":
    main()
