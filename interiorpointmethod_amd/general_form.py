"""General-form front end of the reference, in front of the HIP solver.

    min c^T x   s.t.  Aineq x <= bineq,  Aeq x = beq,  lb <= x <= ub

Same names and argument meaning as the reference (payakorn/InteriorPointMethod):

    get_Abc(c, Aeq, beq, Aineq, bineq, lb, ub, options)   -> (A, b, c, bound)        main.py:818-965
    add_bound_into_matrix(A, b, c, bound)                 -> (A, b, c, (None, None), constant)   main.py:968-1060
    new_interior_sparse(c, Aeq, beq, Aineq, bineq, lb, ub, tol)  -> objective        main.py:1081-1245
    create_problem_from_mps_matlab(name)                  -> (c, Aineq, bineq, Aeq, beq, lb, ub)  sparse_interior.py:290-314

The conversion is host-side data plumbing (scipy.sparse / numpy); the solve is libipm_hip's loop with the
reference's settings for this driver: e1 = e2 = tol, e3 = 1e-6 (main.py:1088-1090), start x = y = s = 1, at most 999
iterations (k starts at 1, main.py:1122, :1127).

Where the reference's code is unfinished the behaviour here is the mathematically intended one, and says so:
  * get_Abc(options="no-bound") with a dense inequality-only problem returns three values in the reference
    (main.py:958); here always four.
  * options="no-bound" discards every FINITE lower bound, zero or not (main.py:896-899): kept, because that is the
    reference's contract for this option; new_interior_sparse() below shifts nonzero lower bounds instead.
  * add_bound_into_matrix: the reference's lower-bound shift has b + A lb (main.py:1047; the substitution x = x' + lb
    gives b - A lb) and its two-sided branch refers to an undefined name (main.py:1053): here b - A lb and a working
    two-sided branch.  `constant` keeps the reference's sign: objective of the original problem = c'^T x' - constant.
  * the reference's corrector takes a dual step of 1 in this driver because a NameError is swallowed
    (main.py:450-455); the device loop uses the ratio test of main.py:604-626 for both steps.
Problems with lb = 0 (all of Netlib's LPs whose reference run converges) are unaffected by the first three.
"""
from __future__ import annotations

import os

import numpy as np
from scipy import sparse

from . import solver as _solver


def _vec(v, name=None, n=None):
    if v is None:
        return None
    v = np.asarray(v, dtype=np.float64).reshape(-1, 1)
    if n is not None and v.shape[0] != n:
        raise ValueError("%s has length %d, expected %d" % (name, v.shape[0], n))
    return v


def _stack_rows(Aineq, Aeq):
    """[[Aineq, I], [Aeq, 0]] (inequality rows first, one slack column per inequality row)."""
    sp = sparse.issparse(Aineq) or (Aeq is not None and sparse.issparse(Aeq))
    mi = Aineq.shape[0]
    if sp:
        top = sparse.hstack([sparse.csc_matrix(Aineq), sparse.identity(mi, format="csc")], format="csc")
        if Aeq is None:
            return top
        bot = sparse.hstack([sparse.csc_matrix(Aeq), sparse.csc_matrix((Aeq.shape[0], mi))], format="csc")
        return sparse.vstack([top, bot], format="csc")
    top = np.hstack([np.asarray(Aineq, dtype=np.float64), np.eye(mi)])
    if Aeq is None:
        return top
    Aeq = np.asarray(Aeq, dtype=np.float64)
    return np.vstack([top, np.hstack([Aeq, np.zeros((Aeq.shape[0], mi))])])


def get_Abc(c, Aeq=None, beq=None, Aineq=None, bineq=None, lb=None, ub=None, options="bound"):
    """Standard form A x = b, x >= 0 (+ a residual `bound`) of a general-form LP; see the module docstring.

    options="bound": a lower bound vector of zeros and an upper bound vector of +inf are dropped (-> None);
    options="no-bound": every finite lower bound is dropped, a -inf lower bound is an error.
    bound is None when nothing is left, else (lb or None, ub or None), padded for the slack columns
    (0 below, +inf above)."""
    if options not in ("bound", "no-bound"):
        raise ValueError('options must be "bound" or "no-bound"')
    if Aeq is None and Aineq is None:
        raise ValueError("get_Abc needs Aeq and/or Aineq")
    c = _vec(c)
    n = c.shape[0]
    lb = np.zeros((n, 1)) if lb is None else _vec(lb, "lb", n)
    ub = np.full((n, 1), np.inf) if ub is None else _vec(ub, "ub", n)
    mi = 0 if Aineq is None else Aineq.shape[0]
    if options == "bound":
        lb_out = None if np.count_nonzero(lb) == 0 else np.vstack([lb, np.zeros((mi, 1))])
    else:
        if not np.all(lb > -np.inf):
            raise ValueError("there are -inf in lower bound")           # the reference raises a str here (TypeError)
        lb_out = None
    ub_out = None if np.all(np.isinf(ub)) else np.vstack([ub, np.full((mi, 1), np.inf)])
    bound = None if (lb_out is None and ub_out is None) else (lb_out, ub_out)
    if Aineq is None:
        return Aeq, beq, c, bound                                          # already equality form: objects passed through
    A = _stack_rows(Aineq, Aeq)
    b = _vec(bineq) if Aeq is None else np.vstack([_vec(bineq), _vec(beq)])
    return A, b, np.vstack([c, np.zeros((mi, 1))]), bound


def add_bound_into_matrix(A, b, c, bound):
    """Fold (lb, ub) into the equality system: x = x' + lb, and one row x'_j + t_j = ub_j - lb_j per finite ub_j.
    Returns (A', b', c', (None, None), constant) with c^T x = c'^T x' - constant."""
    lb, ub = (None, None) if bound is None else bound
    c = _vec(c)
    b = _vec(b)
    n = c.shape[0]
    lb = _vec(lb, "lb", n)
    ub = _vec(ub, "ub", n)
    if lb is not None and np.isinf(lb).any():
        raise ValueError("infinite lower bounds cannot be folded into the matrix")
    constant = 0
    if lb is not None:
        constant = -(c.T @ lb)                                            # (1,1), as the reference returns it
        b = b - A @ lb
        if ub is not None:
            ub = ub - lb
    if ub is None:
        return A, b, c, (None, None), constant
    cols = np.nonzero(np.isfinite(ub).ravel())[0]
    k = len(cols)
    m = A.shape[0]
    sel = sparse.csc_matrix((np.ones(k), (np.arange(k), cols)), shape=(k, n))
    A2 = sparse.vstack([sparse.hstack([sparse.csc_matrix(A), sparse.csc_matrix((m, k))]),
                        sparse.hstack([sel, sparse.identity(k, format="csc")])], format="csc")
    return A2, np.vstack([b, ub[cols]]), np.vstack([c, np.zeros((k, 1))]), (None, None), constant


def standard_form(c, Aeq=None, beq=None, Aineq=None, bineq=None, lb=None, ub=None):
    """What new_interior_sparse() solves: (A, b, c_std, offset) with c^T x = c_std^T x_std + offset."""
    n = np.asarray(c).reshape(-1).shape[0]
    lb_v = np.zeros((n, 1)) if lb is None else _vec(lb, "lb", n)
    shift = np.count_nonzero(lb_v) > 0
    A, b, cs, bound = get_Abc(c, Aeq=Aeq, beq=beq, Aineq=Aineq, bineq=bineq, lb=lb, ub=ub,
                              options="bound" if shift else "no-bound")
    offset = 0.0
    if bound is not None:
        A, b, cs, _, constant = add_bound_into_matrix(A, b, cs, bound)
        offset = -float(np.asarray(constant).reshape(-1)[0]) if np.ndim(constant) else -float(constant)
    return A, _vec(b), _vec(cs), offset


def new_interior_sparse(c, Aeq=None, beq=None, Aineq=None, bineq=None, lb=None, ub=None, tol=1e-20, device=0,
                        return_info=False, start="reference"):
    """Drop-in for main.py:1081-1245: convert to standard form, run the predictor-corrector loop on the GPU
    (e1 = e2 = tol, e3 = 1e-6, at most 999 iterations, x = y = s = 1), return the objective."""
    A, b, cs, offset = standard_form(c, Aeq=Aeq, beq=beq, Aineq=Aineq, bineq=bineq, lb=lb, ub=ub)
    _, _, _, info = _solver.solve_with_info(A, b, cs, tol=tol, tol_gap=1e-6, max_iter=999, y0=1.0, device=device,
                                            start=start)            # start="mehrotra": optional, not the reference's
    obj = info["objective"]
    if info["status"] == 3:                      # NaN iterate: the reference returns the last finite objective it saw
        obj = info["objective_last_finite"]      # (main.py:1227-1233)
    obj = obj + offset
    return (obj, info) if return_info else obj


def create_problem_from_mps_matlab(name, root="."):
    """Loader of the reference's benchmarks_full/<name>.mat files (a MATLAB struct `data` with f, Aineq, bineq, Aeq,
    beq, lb, ub; sparse_interior.py:290-314): empty blocks become None.  scipy.io.loadmat only parses."""
    from scipy.io import loadmat
    rec = loadmat(os.path.join(root, "benchmarks_full", name + ".mat"))["data"]
    get = lambda k: rec[k][0][0]                                       # noqa: E731
    Aineq, Aeq, bineq, beq = get("Aineq"), get("Aeq"), get("bineq"), get("beq")
    if len(bineq) == 0:
        Aineq, bineq = None, None
    if len(beq) == 0:
        Aeq, beq = None, None
    return get("f"), Aineq, bineq, Aeq, beq, get("lb"), get("ub")
