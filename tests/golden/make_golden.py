#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Run in the build container only (needs /root/reference, read-only):

    python3 tests/golden/make_golden.py [--only kat|e2e|dense|dense_large|qap15|netlib] [--jobs 6]

The reference's Python never travels to the GPU box; only the small .npz files
written here do.  Every number stored in an ``expected_*`` / ``ref_*`` key is
the output of the reference's own functions (main.py / sparse_interior.py);
inputs are the reference's benchmarks/*.mat cast to float64.

The reference loops return only an objective (main.py:815) or None
(main.py:757), so the loop is re-composed here from the reference's own step
functions in the exact order of main.py:780-807 / :725-751 to capture (x,y,s).
"""
import argparse
import contextlib
import io
import os
import sys
import time
import warnings

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
from scipy import sparse  # noqa: E402

# the 26 problems on which the verbatim reference loop converges
PARITY_SET = [
    "AFIRO", "BANDM", "DEGEN2", "E226", "FIT1P", "GROW15", "GROW22", "GROW7", "KB2",
    "MAROS-R7", "SC105", "SC205", "SC50A", "SC50B", "SCSD1", "SCSD6", "SCSD8", "SCTAP1",
    "SCTAP2", "SCTAP3", "SHARE2B", "STOCFOR1", "STOCFOR2", "STOCFOR3", "TRUSS", "WOODW",
]


def _import_reference():
    os.chdir(REF)                      # loaders use relative paths (sparse_interior.py:157)
    sys.path.insert(0, REF)
    import main as ref_main            # noqa: E402
    import sparse_interior as ref_si   # noqa: E402
    return ref_main, ref_si


def _quiet():
    return contextlib.redirect_stdout(io.StringIO())


def load_f64(ref_si, name):
    A, b, c, cTlb = ref_si.create_problem_from_mps(name)       # sparse_interior.py:211-216
    A = sparse.csc_matrix(A, dtype=np.float64)
    A.sum_duplicates()
    A.sort_indices()
    return A, np.asarray(b, dtype=np.float64), np.asarray(c, dtype=np.float64), float(cTlb)


def ref_step_sparse(rm, A, b, c, x, y, s, pred_method="full"):
    """One iteration composed of the reference's own functions, order of main.py:783-805."""
    out = {}
    dxa, dya, dsa = rm.direction_predicted_sparse(A, b, c, x, y, s, method=pred_method)
    ap, ad = rm.predicted_stepsize(dxa, dya, dsa, x, s)
    mu_aff, mu_k, sigma = rm.duality_gap(A, x, y, s, dxa, dya, dsa)
    dx, dy, ds = rm.direction_corrected_sparse(A, b, c, x, y, s, dxa, dya, dsa)
    fap, fad = rm.full_stepsize(x, y, s, dx, dy, ds, dxa, dya, dsa)
    xn, yn, sn = rm.corrected(x, y, s, dx, dy, ds, dxa, dya, dsa)
    out.update(dxa=dxa, dya=dya, dsa=dsa, alpha_aff_p=float(ap), alpha_aff_d=float(ad),
               mu_aff=float(np.asarray(mu_aff).ravel()[0]), mu=float(np.asarray(mu_k).ravel()[0]),
               sigma=float(np.asarray(sigma).ravel()[0]), dx=dx, dy=dy, ds=ds,
               alpha_p=float(fap), alpha_d=float(fad), xn=xn, yn=yn, sn=sn)
    return out


def ref_step_dense(rm, A, b, c, x, y, s):
    """Dense-path iteration, order of main.py:728-748."""
    dxa, dya, dsa = rm.direction_predicted(A, b, c, x, y, s)
    ap, ad = rm.predicted_stepsize(dxa, dya, dsa, x, s)
    mu_aff, mu_k, sigma = rm.duality_gap(A, x, y, s, dxa, dya, dsa)
    dx, dy, ds = rm.direction_corrected(A, b, c, x, y, s, dxa, dya, dsa)
    fap, fad = rm.full_stepsize(x, y, s, dx, dy, ds, dxa, dya, dsa)
    xn, yn, sn = rm.corrected(x, y, s, dx, dy, ds, dxa, dya, dsa)
    return dict(dxa=dxa, dya=dya, dsa=dsa, alpha_aff_p=float(ap), alpha_aff_d=float(ad),
                mu_aff=float(np.asarray(mu_aff).ravel()[0]), mu=float(np.asarray(mu_k).ravel()[0]),
                sigma=float(np.asarray(sigma).ravel()[0]), dx=dx, dy=dy, ds=ds,
                alpha_p=float(fap), alpha_d=float(fad), xn=xn, yn=yn, sn=sn)


# ---------------------------------------------------------------- 1. direction KATs
def make_kat(rm, rsi):
    for name, later in (("AFIRO", (5, 60)), ("SC50A", (3, 20)), ("BANDM", (4, 30))):
        A, b, c, cTlb = load_f64(rsi, name)
        m, n = A.shape
        x, y, s = rsi.initial_vector_sparse(m, n)               # sparse_interior.py:193-200
        store = dict(A_data=A.data, A_indices=A.indices, A_indptr=A.indptr,
                     shape=np.array([m, n]), b=b, c=c, cTlb=cTlb,
                     iters=np.array((0,) + later))
        k = 0
        with _quiet(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            while k <= max(later):
                if k in (0,) + later:
                    st = ref_step_sparse(rm, A, b, c, x, y, s, "full")
                    nx, ny, ns = rm.direction_predicted_sparse(A, b, c, x, y, s, method="normal")
                    pre = "k%d_" % k
                    store[pre + "x"], store[pre + "y"], store[pre + "s"] = x, y, s
                    for key, val in st.items():
                        store[pre + key] = val
                    store[pre + "normal_dxa"], store[pre + "normal_dya"], store[pre + "normal_dsa"] = nx, ny, ns
                    store[pre + "continue"] = bool(rm.check_optimality(A, b, c, x, y, s, 1e-8, 1e-8, 1e-8,
                                                                        options="sparse"))
                    x, y, s = st["xn"], st["yn"], st["sn"]
                else:
                    x, y, s = _advance(rm, A, b, c, x, y, s)
                k += 1
        np.savez_compressed(os.path.join(HERE, "kat_%s.npz" % name), **store)
        print("kat", name, "ok")


def _advance(rm, A, b, c, x, y, s):
    st = ref_step_sparse(rm, A, b, c, x, y, s, "full")
    return st["xn"], st["yn"], st["sn"]


# ---------------------------------------------------------------- 2. end-to-end KATs
def _e2e_one(name):
    rm, rsi = _import_reference()
    A, b, c, cTlb = load_f64(rsi, name)
    m, n = A.shape
    tol = 1e-8
    x, y, s = rsi.initial_vector_sparse(m, n)
    k = 0
    t0 = time.time()
    with _quiet(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        while rm.check_optimality(A, b, c, x, y, s, tol, tol, tol, options="sparse") and k < 5000:
            x, y, s = _advance(rm, A, b, c, x, y, s)              # main.py:783-805
            k += 1
        cont = bool(rm.check_optimality(A, b, c, x, y, s, tol, tol, tol, options="sparse"))
    secs = time.time() - t0
    obj = float(np.sum(x * c))                                     # main.py:815 (before -cTlb)
    rb = A @ x - b
    rc = A.T @ y + s - c
    np.savez_compressed(
        os.path.join(HERE, "e2e_%s.npz" % name),
        name=name, shape=np.array([m, n]), tol=tol, iterations=k, converged=not cont,
        objective=obj, cTlb=cTlb,
        rp=float(np.linalg.norm(rb) / (1 + np.linalg.norm(b))),
        rd=float(np.linalg.norm(rc) / (1 + np.linalg.norm(c))),
        gap=float((x.T @ s)[0, 0]), x=x, y=y, s=s, ref_seconds=secs)
    return name, k, obj, secs


def make_e2e(jobs, names=None):
    import multiprocessing as mp
    names = names or PARITY_SET
    # longest first so the pool's makespan is WOODW's solve
    order = sorted(names, key=lambda n: -{"WOODW": 400, "MAROS-R7": 120, "STOCFOR3": 45, "TRUSS": 40,
                                          "DEGEN2": 15}.get(n, 1))
    with mp.get_context("spawn").Pool(jobs) as pool:
        for name, k, obj, secs in pool.imap_unordered(_e2e_one, order):
            print("e2e %-10s it=%4d obj=%.13e  %.1fs" % (name, k, obj, secs), flush=True)


# ---------------------------------------------------------------- 3. dense KATs
def synthetic_lp(m, n, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    x0 = rng.uniform(0.5, 1.5, (n, 1))
    y0 = rng.standard_normal((m, 1))
    s0 = rng.uniform(0.5, 1.5, (n, 1))
    return A, A @ x0, A.T @ y0 + s0


def make_dense(rm):
    cases = {}
    for nm in ("ex1", "ex2", "ex3"):
        A, b, c = getattr(rm, nm)()                               # main.py:1249-1282
        cases[nm] = rm.convert_to_array(A, b, c)                  # main.py:280-284
    for (m, n) in ((64, 128), (256, 512), (512, 1024)):
        cases["syn_%dx%d" % (m, n)] = synthetic_lp(m, n)
    tol = 1e-8
    for nm, (A, b, c) in cases.items():
        A = np.asarray(A, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        c = np.asarray(c, dtype=np.float64)
        x, y, s = rm.initial_vector(A)                            # main.py:287-302 (y=0)
        k = 0
        store = dict(A=A if A.size <= 64 * 128 else np.zeros(0), b=b, c=c, shape=np.array(A.shape),
                     tol=tol, seed=0)
        with _quiet():
            while rm.check_optimality(A, b, c, x, y, s, tol, tol, tol) and k < 50000:
                st = ref_step_dense(rm, A, b, c, x, y, s)
                if k == 0:
                    for key, val in st.items():
                        store["k0_" + key] = val
                x, y, s = st["xn"], st["yn"], st["sn"]
                k += 1
        store.update(iterations=k, objective=float(np.sum(x * c)), x=x, y=y, s=s,
                     gap=float((x.T @ s)[0, 0]),
                     rp=float(np.linalg.norm(A @ x - b) / (1 + np.linalg.norm(b))),
                     rd=float(np.linalg.norm(A.T @ y + s - c) / (1 + np.linalg.norm(c))))
        np.savez_compressed(os.path.join(HERE, "dense_%s.npz" % nm), **store)
        print("dense %-12s it=%3d obj=%.13e" % (nm, k, store["objective"]), flush=True)


def make_dense_large(rm, sizes=((1024, 2048), (4096, 8192))):
    """Reference-verbatim dense solves above 512 x 1024 (the sizes where the HIP path's grouped triangular solves
    and, at 4096 x 8192, the headline configuration run).  A itself is NOT stored (268 MB at 4096 x 8192: it is
    regenerated from the seed by the same generator); stored are the reference's final (x, y, s), objective,
    iteration count, residuals, the first-step scalars and the objective after every iteration.  4096 x 8192 takes
    ~45-60 minutes (two LAPACK gesv of a 20480^2 matrix per iteration, main.py:185-244) and ~10 GB."""
    tol = 1e-8
    for (m, n) in sizes:
        A, b, c = synthetic_lp(m, n)
        x, y, s = rm.initial_vector(A)                            # main.py:287-302 (y=0)
        k = 0
        store = dict(b=b, c=c, shape=np.array(A.shape), tol=tol, seed=0)
        traj = []
        t0 = time.time()
        with _quiet():
            while rm.check_optimality(A, b, c, x, y, s, tol, tol, tol) and k < 50000:
                st = ref_step_dense(rm, A, b, c, x, y, s)
                if k == 0:
                    for key, val in st.items():
                        store["k0_" + key] = val
                x, y, s = st["xn"], st["yn"], st["sn"]
                k += 1
                traj.append(float(np.sum(x * c)))
                print("dense_large %dx%d k=%d obj=%.13e  %.0fs" % (m, n, k, traj[-1], time.time() - t0),
                      file=sys.stderr, flush=True)
        store.update(iterations=k, objective=float(np.sum(x * c)), x=x, y=y, s=s,
                     objective_after_iteration=np.array(traj),
                     gap=float((x.T @ s)[0, 0]),
                     rp=float(np.linalg.norm(A @ x - b) / (1 + np.linalg.norm(b))),
                     rd=float(np.linalg.norm(A.T @ y + s - c) / (1 + np.linalg.norm(c))),
                     ref_seconds=time.time() - t0)
        np.savez_compressed(os.path.join(HERE, "dense_syn_%dx%d.npz" % (m, n)), **store)
        print("dense %-12s it=%3d obj=%.13e  %.0fs" % ("syn_%dx%d" % (m, n), k, store["objective"],
                                                      store["ref_seconds"]), flush=True)


# ---------------------------------------------------------------- 4. QAP15 direction KAT
def make_qap15(rm, rsi):
    A, b, c, cTlb = load_f64(rsi, "QAP15")
    m, n = A.shape
    x, y, s = rsi.initial_vector_sparse(m, n)
    t0 = time.time()
    with _quiet(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dxa, dya, dsa = rm.direction_predicted_sparse(A, b, c, x, y, s, method="normal")  # main.py:221-229
        ap, ad = rm.predicted_stepsize(dxa, dya, dsa, x, s)
    np.savez_compressed(os.path.join(HERE, "kat_QAP15_normal_k0.npz"), shape=np.array([m, n]),
                        dxa=dxa, dya=dya, dsa=dsa, alpha_aff_p=float(ap), alpha_aff_d=float(ad),
                        netlib_optimum=1.0409940410e3, ref_seconds=time.time() - t0)
    print("qap15 ok %.1fs" % (time.time() - t0))


# ---------------------------------------------------------------- 5. Netlib inputs (data only)
def make_netlib(rsi):
    """benchmarks/*.mat -> tests/golden/netlib/<NAME>.npz (CSC triplets, b, c, cTlb; float64).

    Pure input data: the LP instances, no reference code.  The eight files with
    +-Inf/NaN in b are kept and flagged ``valid=False``.
    """
    from scipy.io import loadmat
    out = os.path.join(HERE, "netlib")
    os.makedirs(out, exist_ok=True)
    names = sorted(f[:-4] for f in os.listdir(os.path.join(REF, "benchmarks")) if f.endswith(".mat"))
    for name in names:
        d = loadmat(os.path.join(REF, "benchmarks", name + ".mat"))   # keys: sparse_interior.py:157-167
        i = d["A"]["i"][0][0][0].astype(np.int64)
        j = d["A"]["j"][0][0][0].astype(np.int64)
        k = d["A"]["k"][0][0][0].astype(np.float64)
        m = int(d["num_constraints"][0][0])
        n = int(d["num_variables"][0][0])
        A = sparse.csc_matrix((k, (i, j)), shape=(max(m, int(i.max()) + 1), max(n, int(j.max()) + 1)))
        A.sum_duplicates()
        A.sort_indices()
        b = np.asarray(d["b"], dtype=np.float64).reshape(-1)
        c = np.asarray(d["f"], dtype=np.float64).reshape(-1)
        cTlb = float(d["cTlb"][0][0])
        valid = bool(np.all(np.isfinite(b)) and np.all(np.isfinite(c)) and np.isfinite(cTlb)
                     and np.all(np.isfinite(A.data)))
        np.savez_compressed(os.path.join(out, name + ".npz"), indptr=A.indptr.astype(np.int32),
                            indices=A.indices.astype(np.int32), data=A.data, shape=np.array(A.shape),
                            b=b, c=c, cTlb=cTlb, valid=valid)
    print("netlib: %d files" % len(names))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--names", default="")
    args = ap.parse_args()
    if args.only in ("all", "e2e"):
        make_e2e(args.jobs, [n for n in args.names.split(",") if n] or None)
    if args.only != "e2e":
        rm, rsi = _import_reference()
        if args.only in ("all", "kat"):
            make_kat(rm, rsi)
        if args.only in ("all", "dense"):
            make_dense(rm)
        if args.only == "dense_large":          # not part of "all": ~1 hour
            make_dense_large(rm, [tuple(int(v) for v in t.split("x")) for t in args.names.split(",") if t]
                             or ((1024, 2048), (4096, 8192)))
        if args.only in ("all", "netlib"):
            make_netlib(rsi)
        if args.only in ("all", "qap15"):
            make_qap15(rm, rsi)
