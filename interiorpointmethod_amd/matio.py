"""LP loaders: the reference's Netlib ``.mat`` layout and this repo's ``.npz`` fixtures.

``create_problem_from_mps`` mirrors sparse_interior.py:211-216 (keys f, b, cTlb, A{i,j,k},
num_variables, num_constraints; sparse_interior.py:157-167) but casts to float64 (the files
hold int16/uint8) and passes the shape explicitly instead of inferring it from the
largest index (sparse_interior.py:215).
"""
from __future__ import annotations

import os

import numpy as np
from scipy import sparse


def create_problem_from_mps(name, root="benchmarks"):
    """-> (A csc float64 (m,n), b (m,1), c (n,1), cTlb float) from ``<root>/<name>.mat``."""
    from scipy.io import loadmat
    d = loadmat(os.path.join(root, "%s.mat" % name))
    i = d["A"]["i"][0][0][0].astype(np.int64)
    j = d["A"]["j"][0][0][0].astype(np.int64)
    k = d["A"]["k"][0][0][0].astype(np.float64)
    m = max(int(d["num_constraints"][0][0]), int(i.max()) + 1)
    n = max(int(d["num_variables"][0][0]), int(j.max()) + 1)
    A = sparse.csc_matrix((k, (i, j)), shape=(m, n))
    A.sum_duplicates()
    b = np.asarray(d["b"], dtype=np.float64).reshape(-1, 1)
    c = np.asarray(d["f"], dtype=np.float64).reshape(-1, 1)
    return A, b, c, float(d["cTlb"][0][0])


def load_npz_problem(path):
    """-> (A csc, b (m,1), c (n,1), cTlb, valid) from a tests/golden/netlib/*.npz fixture."""
    z = np.load(path)
    m, n = (int(v) for v in z["shape"])
    A = sparse.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(m, n))
    return A, z["b"].reshape(-1, 1), z["c"].reshape(-1, 1), float(z["cTlb"]), bool(z["valid"])


def is_valid_problem(A, b, c):
    """False for the eight Netlib files whose b/c carry +-Inf/NaN."""
    data = A.data if sparse.issparse(A) else np.asarray(A)
    return bool(np.all(np.isfinite(data)) and np.all(np.isfinite(b)) and np.all(np.isfinite(c)))
