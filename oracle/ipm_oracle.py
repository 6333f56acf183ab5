"""CPU oracle for the Newton/KKT hot path of payakorn/InteriorPointMethod.

TEST INFRASTRUCTURE ONLY.  This module is a NumPy/SciPy restatement of the
reference's Mehrotra predictor-corrector iteration.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker, never as the thing shipped or measured.  The
product path (``interiorpointmethod_amd``) never imports it and fails loudly
when the HIP library is missing.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the reference's own functions in
the build container (``tests/golden/make_golden.py``; reference = main.py /
sparse_interior.py of /root/reference).  The reference's own test-suite pins no
numeric value on this path (test.py checks shapes only), so those generated
vectors plus the known-answer optima in main.py:1253/:1261 and
benchmarks/readme.txt:83-182 are the pins.

Two formulations are restated:

* ``method="full"``   -- the unreduced (m+2n) KKT system solved by LU, i.e. what
  the reference loops ``interior`` (main.py:707-757) and ``interior_sparse``
  (main.py:760-815) actually execute.  This is the "reference algorithm" leg.
* ``method="normal"`` -- the normal-equations form ``A D^2 A^T`` of
  main.py:221-229 with ONE Cholesky factorization reused by predictor and
  corrector and a LIPSOL-style pivot guard.  This is the algorithm the HIP path
  implements; SURVEY.md section 3.5 derives the corrector in this form.

All vectors are (len, 1) float64 column arrays, exactly like the reference.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
from scipy import sparse
from scipy.sparse.linalg import spsolve

ETA = 0.91            # main.py:607 (hard-coded damping of the corrector step)
PIVOT_GUARD_EPS = 1e-30   # pivot <= eps * max(diag B)  ->  pivot := PIVOT_GUARD_BIG
PIVOT_GUARD_BIG = 1e64

STATUS_OK = 0
STATUS_MAX_ITER = 1
STATUS_NAN = 2


def _col(v):
    v = np.asarray(v, dtype=np.float64)
    return v.reshape(-1, 1)


def as_float64_problem(A, b, c):
    """Cast (A, b, c) to float64 (the .mat files carry int16/uint8)."""
    if sparse.issparse(A):
        A = sparse.csc_matrix(A, dtype=np.float64)
    else:
        A = np.ascontiguousarray(np.asarray(A, dtype=np.float64))
    return A, _col(b), _col(c)


# --------------------------------------------------------------------------
# start points                                  sparse_interior.py:193-200 / main.py:287-302
# --------------------------------------------------------------------------
def initial_point(m, n, y0=1.0):
    """x = s = 1, y = y0 (1.0 on the sparse path, 0.0 on the dense path)."""
    return np.ones((n, 1)), np.full((m, 1), float(y0)), np.ones((n, 1))


# --------------------------------------------------------------------------
# residuals + stop test                          main.py:66-73, 162-173
# --------------------------------------------------------------------------
def residuals(A, b, c, x, y, s):
    """(r_c, r_b, r3) = (A^T y + s - c, A x - b, x*s)   -- main.py:66-73."""
    rb = A @ x - b
    rc = A.T @ y + s - c
    return rc, rb, x * s


def check_optimality(A, b, c, x, y, s, e1, e2, e3):
    """True while the loop must CONTINUE -- main.py:162-173.

    primal: e1 (1+||b||) < ||Ax-b||,  dual: e2 (1+||c||) < ||A^T y + s - c||,
    gap: e3 < x^T s  (absolute complementarity, not mu).
    """
    rc, rb, _ = residuals(A, b, c, x, y, s)
    primal = e1 * (1.0 + np.linalg.norm(b)) < np.linalg.norm(rb)
    dual = e2 * (1.0 + np.linalg.norm(c)) < np.linalg.norm(rc)
    gap = e3 < float((x.T @ s)[0, 0])
    return bool(primal or dual or gap)


# --------------------------------------------------------------------------
# step lengths / centering                       main.py:305-322, 562-601, 604-626
# --------------------------------------------------------------------------
def ratio_test(v, dv):
    """min(1, min_{dv_i<0} -v_i/dv_i)   -- main.py:308-319 and :615-617."""
    neg = dv < 0
    if not neg.any():
        return 1.0
    return float(min(1.0, np.min(-v[neg] / dv[neg])))


def predicted_stepsize(dx, ds, x, s):
    return ratio_test(x, dx), ratio_test(s, ds)


def centering(x, s, dx_aff, ds_aff):
    """(mu_aff, mu, sigma=(mu_aff/mu)^3)   -- main.py:588-601."""
    n = x.shape[0]
    ap, ad = predicted_stepsize(dx_aff, ds_aff, x, s)
    mu_aff = float(((x + ap * dx_aff).T @ (s + ad * ds_aff))[0, 0]) / n
    mu = float((x.T @ s)[0, 0]) / n
    return mu_aff, mu, (mu_aff / mu) ** 3


def full_stepsize(dx, ds, x, s, eta=ETA):
    """alpha = min(1, eta * min(1, ratio))   -- main.py:604-626 (so alpha <= eta)."""
    return min(1.0, eta * ratio_test(x, dx)), min(1.0, eta * ratio_test(s, ds))


# --------------------------------------------------------------------------
# full-KKT direction (reference algorithm)       main.py:13-21, 185-212, 232-269;
#                                                sparse_interior.py:57-96
# --------------------------------------------------------------------------
def kkt_matrix(A, x, s):
    """[[0, A^T, I], [A, 0, 0], [S, 0, X]] of order m+2n."""
    m, n = A.shape
    if sparse.issparse(A):
        I = sparse.identity(n, format="csc")
        S = sparse.diags(s.ravel())
        X = sparse.diags(x.ravel())
        return sparse.bmat([[None, A.T, I], [A, None, None], [S, None, X]], format="csc")
    K = np.zeros((m + 2 * n, m + 2 * n))
    K[:n, n:n + m] = A.T
    K[:n, n + m:] = np.eye(n)
    K[n:n + m, :n] = A
    K[n + m:, :n] = np.diagflat(s)
    K[n + m:, n + m:] = np.diagflat(x)
    return K


def direction_full(A, rc, rb, r3, x, s):
    """Solve the unreduced KKT system for rhs [-rc; -rb; -r3] (main.py:105-108, 150-158)."""
    m, n = A.shape
    K = kkt_matrix(A, x, s)
    rhs = np.vstack([-rc, -rb, -r3])
    if sparse.issparse(K):
        sol = spsolve(K, rhs).reshape(-1, 1)          # SuperLU, main.py:180
    else:
        sol = np.linalg.solve(K, rhs)                 # LAPACK gesv, main.py:178
    return sol[:n], sol[n:n + m], sol[n + m:]


# --------------------------------------------------------------------------
# normal-equations direction (what the HIP path computes)   main.py:221-229
# --------------------------------------------------------------------------
def form_normal_matrix(A, d):
    """B = A diag(d) A^T as a dense m x m array   -- main.py:223-224."""
    dd = np.asarray(d).ravel()
    if sparse.issparse(A):
        AD = A @ sparse.diags(dd)
        return np.asarray((AD @ A.T).todense())
    return (A * dd) @ A.T


def guarded_cholesky(B, eps=PIVOT_GUARD_EPS, big=PIVOT_GUARD_BIG):
    """Lower Cholesky factor with a LIPSOL-style pivot guard.

    A pivot p <= eps * max(diag B) (or non-finite/non-positive) is replaced by
    ``big`` so that the corresponding solution component is driven to ~0.
    Returns (L, n_fixed).  Tries LAPACK first; the guarded right-looking loop
    only runs when plain Cholesky breaks down.
    """
    m = B.shape[0]
    thresh = eps * float(np.max(np.diag(B))) if m else 0.0
    try:
        L = sla.cholesky(B, lower=True, check_finite=False)
        if np.all(np.isfinite(L)) and np.min(np.diag(L)) ** 2 > thresh:
            return L, 0
    except sla.LinAlgError:
        pass
    W = np.array(B, dtype=np.float64, copy=True)
    fixed = 0
    for j in range(m):
        p = W[j, j]
        if not (p > thresh):                           # also catches NaN
            p = big
            fixed += 1
        ljj = np.sqrt(p)
        W[j, j] = ljj
        if j + 1 < m:
            W[j + 1:, j] /= ljj
            col = W[j + 1:, j]
            W[j + 1:, j + 1:] -= np.outer(col, col)    # symmetric right-looking update
    return np.tril(W), fixed


def cholesky_solve(L, rhs):
    z = sla.solve_triangular(L, rhs, lower=True, check_finite=False)
    return sla.solve_triangular(L.T, z, lower=False, check_finite=False)


def direction_normal(A, rc, rb, r3, x, s, L=None):
    """Normal-equations Newton direction for complementarity residual r3.

    d = x/s, t = rc - r3/x, B dy = -rb - A(d*t), dx = d*(A^T dy) + d*t,
    ds = -s*dx/x - r3/x     (main.py:223-228).
    Returns (dx, dy, ds, L, n_fixed); pass L back in to reuse the factor.
    """
    d = x / s
    t = rc - r3 / x
    v = d * t
    fixed = 0
    if L is None:
        L, fixed = guarded_cholesky(form_normal_matrix(A, d))
    rhs = -rb - A @ v
    dy = cholesky_solve(L, rhs)
    dx = d * (A.T @ dy) + v
    ds = -s * dx / x - r3 / x
    return dx, dy, ds, L, fixed


# --------------------------------------------------------------------------
# one Mehrotra iteration + driver                main.py:725-751, 780-807
# --------------------------------------------------------------------------
def iterate(A, b, c, x, y, s, method="normal", eta=ETA):
    """One predictor-corrector step. Returns new (x,y,s) and info."""
    n = x.shape[0]
    rc, rb, r3 = residuals(A, b, c, x, y, s)
    L = None
    fixed = 0
    if method == "normal":
        dxa, dya, dsa, L, fixed = direction_normal(A, rc, rb, r3, x, s)
    else:
        dxa, dya, dsa = direction_full(A, rc, rb, r3, x, s)
    mu_aff, mu, sigma = centering(x, s, dxa, dsa)
    r3c = x * s + dxa * dsa - sigma * mu * np.ones((n, 1))      # main.py:150-152
    if method == "normal":
        dx, dy, ds, _, _ = direction_normal(A, rc, rb, r3c, x, s, L=L)
    else:
        dx, dy, ds = direction_full(A, rc, rb, r3c, x, s)
    ap, ad = full_stepsize(dx, ds, x, s, eta)
    xn = x + ap * dx                                             # main.py:694-696
    yn = y + ad * dy
    sn = s + ad * ds
    info = dict(mu=mu, mu_aff=mu_aff, sigma=sigma, alpha_p=ap, alpha_d=ad,
                pivots_fixed=fixed, dxa=dxa, dya=dya, dsa=dsa, dx=dx, dy=dy, ds=ds)
    return xn, yn, sn, info


def solve(A, b, c, tol=1e-8, max_iter=5000, y0=1.0, method="normal", eta=ETA,
          tol_gap=None, callback=None):
    """Restatement of the reference loops (main.py:760-815 with y0=1; :707-757 with y0=0).

    Stop test FIRST, then a full predictor-corrector step; e1=e2=tol, e3=tol
    (main.py:772-774) unless ``tol_gap`` is given (new_interior_sparse uses 1e-6,
    main.py:1088-1090).  Returns (x, y, s, info).
    """
    A, b, c = as_float64_problem(A, b, c)
    m, n = A.shape
    e3 = tol if tol_gap is None else tol_gap
    x, y, s = initial_point(m, n, y0)
    k = 0
    status = STATUS_OK
    fixed_total = 0
    while check_optimality(A, b, c, x, y, s, tol, tol, e3):
        if k >= max_iter:
            status = STATUS_MAX_ITER
            break
        xn, yn, sn, it = iterate(A, b, c, x, y, s, method=method, eta=eta)
        if not (np.all(np.isfinite(xn)) and np.all(np.isfinite(yn)) and np.all(np.isfinite(sn))):
            status = STATUS_NAN                      # main.py:1141-1148 NaN probe
            break
        x, y, s = xn, yn, sn
        fixed_total += it["pivots_fixed"]
        k += 1
        if callback is not None:
            callback(k, x, y, s, it)
    rc, rb, _ = residuals(A, b, c, x, y, s)
    info = dict(
        iterations=k, status=status,
        objective=float((c.T @ x)[0, 0]),             # sum(x*c), main.py:815
        rp=float(np.linalg.norm(rb) / (1.0 + np.linalg.norm(b))),
        rd=float(np.linalg.norm(rc) / (1.0 + np.linalg.norm(c))),
        gap=float((x.T @ s)[0, 0]),
        pivots_fixed=fixed_total,
    )
    return x, y, s, info


# --------------------------------------------------------------------------
# synthetic dense LP of SURVEY 8(d) (the bench workload)
# --------------------------------------------------------------------------
def synthetic_lp(m, n, seed=0):
    """Strictly feasible dense LP: draws in the order fixed by SURVEY 8(d)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    x0 = rng.uniform(0.5, 1.5, (n, 1))
    y0 = rng.standard_normal((m, 1))
    s0 = rng.uniform(0.5, 1.5, (n, 1))
    b = A @ x0
    c = A.T @ y0 + s0
    return A, b, c


def flops_per_iteration(m, n, nnz_col_sq=None):
    """Algorithmic flops of one iteration, SURVEY 8(d): m^2 n + m^3/3 + 4 m^2 + 12 m n."""
    form = float(m) * m * n if nnz_col_sq is None else float(nnz_col_sq)
    return form + m ** 3 / 3.0 + 4.0 * m * m + 12.0 * m * n
