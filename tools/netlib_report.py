#!/usr/bin/env python3
"""The reference driver's results table (script.py:139-198, written to conclusion1.txt) with the HIP path in the
"Interi" columns:

    Name | Interi time | Scipy time | Obj fun | Interi | Scipy

--set general (default): the general-form inputs the reference driver actually loops over (benchmarks_full/, here the
committed fixtures tests/golden/general/*.npz) through new_interior_sparse(tol=1e-6) exactly as script.py:171-173
calls it; "Scipy" is scipy.optimize.linprog(method="interior-point") on the host as in script.py:155-164 (falls back to
"highs-ipm" where SciPy no longer ships it); "Obj fun" is the Netlib optimum the reference carries (main.py:1417-1516).
--set standard: the standard-form benchmarks/ fixtures through interior_sparse's loop (tol=1e-8, cap 300); "Obj fun" is
the reference's own objective where its verbatim loop converges (tests/golden/e2e_*.npz).
--cpu adds a "CPU port time" column: the oracle (NumPy restatement of the same algorithm) on the host, small LPs only.

    python tools/netlib_report.py [--set general|standard] [--out conclusion_gpu.txt] [--start mehrotra] [--cpu] [NAME ...]

Same row format as the reference ("{0:17s} {2:17.2f} {3:>20.2f} {1:20.2f} {4:20.2f} {5:20.2f}"): a file written here
diffs against the reference's conclusion1.txt.
"""
import argparse
import glob
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")

HEADER_FMT = "{0:17s} {2:>17s} {3:>20s} {1:>20s} {4:>20s} {5:>20s}\r\n"          # script.py:141-145
ROW_FMT = "{0:17s} {2:17.2f} {3:>20.2f} {1:20.2f} {4:20.2f} {5:20.2f}\r\n"        # script.py:187-196


def header(extra=()):
    line = HEADER_FMT.format("Name", "Obj fun", "Interi time", "Scipy time", "Interi", "Scipy")
    return line[:-2] + "".join(" {0:>16s}".format(e) for e in extra) + "\r\n"


def row(name, obj_netlib, t_interi, t_scipy, obj_interi, obj_scipy, extra=()):
    line = ROW_FMT.format(name, obj_netlib, t_interi, t_scipy, obj_interi, obj_scipy)
    return line[:-2] + "".join(" {0:>16s}".format(e) for e in extra) + "\r\n"


def _mat(z, prefix):
    from scipy import sparse
    if prefix + "_none" in z.files:
        return None
    return sparse.csc_matrix((z[prefix + "_data"], z[prefix + "_indices"], z[prefix + "_indptr"]),
                             shape=tuple(int(v) for v in z[prefix + "_shape"]))


def load_general(name):
    """(c, Aineq, bineq, Aeq, beq, lb, ub, netlib optimum) of a general-form fixture."""
    z = np.load(os.path.join(G, "general", name + ".npz"))
    return (z["c"], _mat(z, "Aineq"), z["bineq"] if "bineq" in z.files else None, _mat(z, "Aeq"),
            z["beq"] if "beq" in z.files else None, z["lb"], z["ub"], float(z["netlib_optimum"]))


def scipy_reference(c, Aineq, bineq, Aeq, beq, lb, ub):
    """script.py:155-164: scipy.optimize.linprog on the host -> (objective, seconds)."""
    from scipy.optimize import linprog
    lb = np.asarray(lb, dtype=np.float64).ravel(); ub = np.asarray(ub, dtype=np.float64).ravel()
    bounds = [(None if not np.isfinite(l) else float(l), None if not np.isfinite(u) else float(u)) for l, u in zip(lb, ub)]
    t0 = time.time()
    res = None
    for method in ("interior-point", "highs-ipm"):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = linprog(c=np.asarray(c, dtype=np.float64).ravel(), A_ub=Aineq,
                              b_ub=None if bineq is None else np.asarray(bineq, dtype=np.float64).ravel(), A_eq=Aeq,
                              b_eq=None if beq is None else np.asarray(beq, dtype=np.float64).ravel(), bounds=bounds,
                              method=method)
            break
        except ValueError:
            continue
    return (float(res.fun) if res is not None and res.fun is not None else float("nan")), time.time() - t0


def run_general(names, solve_fn, with_scipy=True, with_cpu=False):
    """Rows of the table for general-form fixtures.  solve_fn(c, Aineq, bineq, Aeq, beq, lb, ub) -> objective."""
    rows = []
    for nm in names:
        c, Aineq, bineq, Aeq, beq, lb, ub, opt = load_general(nm)
        obj_s, t_s = scipy_reference(c, Aineq, bineq, Aeq, beq, lb, ub) if with_scipy else (float("nan"), float("nan"))
        t0 = time.time()
        obj = solve_fn(c, Aineq, bineq, Aeq, beq, lb, ub)
        t_i = time.time() - t0
        extra = ()
        if with_cpu:
            extra = ("-",)
            if c.shape[0] <= 1200:
                from oracle import ipm_oracle as O
                from interiorpointmethod_amd import general_form as GF
                A, b, cs, off = GF.standard_form(c, Aeq=Aeq, beq=beq, Aineq=Aineq, bineq=bineq, lb=lb, ub=ub)
                t0 = time.time()
                O.solve(A, b, cs, tol=1e-6, tol_gap=1e-6, y0=1.0, method="normal", max_iter=999)
                extra = ("%.2f" % (time.time() - t0),)
        rows.append((nm, opt, t_i, t_s, obj, obj_s, extra))
    return rows


def write_table(rows, path, extra_header=()):
    with open(path, "w", newline="") as fh:
        fh.write(header(extra_header))
        for nm, opt, t_i, t_s, obj, obj_s, extra in rows:
            fh.write(row(nm, opt, t_i, t_s, obj, obj_s, extra))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--set", default="general", choices=["general", "standard"])
    ap.add_argument("--out", default="conclusion_gpu.txt")
    ap.add_argument("--start", default="reference", choices=["reference", "mehrotra"])
    ap.add_argument("--no-scipy", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="add the CPU oracle's wall time (LPs with <= 1200 variables)")
    ap.add_argument("--max-iter", type=int, default=300)
    args = ap.parse_args()
    import interiorpointmethod_amd as ipm
    if args.set == "general":
        from interiorpointmethod_amd import general_form as GF
        names = args.names or sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(G, "general", "*.npz")))
        solve = lambda c, Aineq, bineq, Aeq, beq, lb, ub: GF.new_interior_sparse(            # noqa: E731
            c=c, Aineq=Aineq, bineq=bineq, Aeq=Aeq, beq=beq, lb=lb, ub=ub, tol=1e-6, start=args.start)   # script.py:171-173
        rows = run_general(names, solve, with_scipy=not args.no_scipy, with_cpu=args.cpu)
        write_table(rows, args.out, ("CPU port time",) if args.cpu else ())
    else:
        from interiorpointmethod_amd.matio import load_npz_problem
        names = args.names or sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(G, "netlib", "*.npz")))
        rows = []
        for nm in names:
            A, b, c, cTlb, valid = load_npz_problem(os.path.join(G, "netlib", nm + ".npz"))
            e2e = os.path.join(G, "e2e_%s.npz" % nm)
            ref = float(np.load(e2e)["objective"]) if os.path.exists(e2e) else float("nan")
            if not valid:
                rows.append((nm, ref, float("nan"), float("nan"), float("nan"), float("nan"), ("invalid input",)))
                continue
            t0 = time.time()
            _, _, _, info = ipm.solve_with_info(A, b, c, tol=1e-8, max_iter=args.max_iter, start=args.start)
            rows.append((nm, ref, time.time() - t0, float("nan"), info["objective"], float("nan"),
                         ("%d its %s" % (info["iterations"], info["status_name"]),)))
        write_table(rows, args.out, ("iterations/status",))
    print(open(args.out).read())


if __name__ == "__main__":
    main()
