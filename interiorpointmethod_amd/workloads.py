"""Benchmark workloads of BASELINE.json (inputs only; no solver code)."""
from __future__ import annotations

import numpy as np


def synthetic_lp(m, n, seed=0):
    """Strictly primal-dual feasible dense LP of SURVEY.md 8(d).

    PCG64 draws in this order: A ~ N(0,1) (m,n); x0 ~ U(0.5,1.5) (n,1); y0 ~ N(0,1) (m,1);
    s0 ~ U(0.5,1.5) (n,1); b = A x0; c = A^T y0 + s0.  Pin: A[0,0] = 0.1257302210933933.
    """
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n))
    x0 = rng.uniform(0.5, 1.5, (n, 1))
    y0 = rng.standard_normal((m, 1))
    s0 = rng.uniform(0.5, 1.5, (n, 1))
    return A, A @ x0, A.T @ y0 + s0


def flops_per_iteration(m, n, nnz_col_sq=None):
    """Algorithmic flops of one iteration: m^2 n + m^3/3 + 4 m^2 + 12 m n;
    for a sparse A the contraction term is sum_j nnz(A[:,j])^2."""
    form = float(m) * m * n if nnz_col_sq is None else float(nnz_col_sq)
    return form + m ** 3 / 3.0 + 4.0 * m * m + 12.0 * m * n
