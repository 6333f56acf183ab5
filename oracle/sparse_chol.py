"""ctypes front of oracle/sparse_chol_oracle.cpp -- TEST INFRASTRUCTURE ONLY (tests/, smoke): the sequential CPU
restatement of the multifrontal sparse Cholesky of csrc/sparse_chol.h.  Built by __graft_entry__.build()."""
import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsparse_chol_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
    return _lib


def factor_solve(A, d, rhs, eps=1e-30, big=1e64, shift_rel=0.0, wcap=32, lds=7680, want_factor=True):
    """Sparse Cholesky of A diag(d) A^T in the minimum-degree row order + one solve.  Returns dict(perm, L (dense, permuted
    order, or None), z (caller's row order), fixed, stats)."""
    A = sp.csc_matrix(A, dtype=np.float64)
    A.sort_indices()
    m, n = A.shape
    pi, pd = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    perm = np.zeros(m, dtype=np.int32)
    L = np.zeros((m, m)) if want_factor else None
    z = np.zeros(m)
    nf = C.c_int(0)
    st = np.zeros(8)
    ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
    ii = np.ascontiguousarray(A.indices, dtype=np.int32)
    dv = np.ascontiguousarray(A.data, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    rc = load().spchol_oracle(m, n, ip.ctypes.data_as(pi), ii.ctypes.data_as(pi), dv.ctypes.data_as(pd), d.ctypes.data_as(pd),
                              rhs.ctypes.data_as(pd), C.c_double(eps), C.c_double(big), C.c_double(shift_rel), wcap, lds,
                              perm.ctypes.data_as(pi), L.ctypes.data_as(pd) if want_factor else None, z.ctypes.data_as(pd),
                              C.byref(nf), st.ctypes.data_as(pd))
    if rc:
        raise RuntimeError("spchol_oracle failed with code %d" % rc)
    keys = ("panels", "height", "widest_front", "factor_entries", "update_entries", "flops", "fan_in_nodes", "max_children")
    return dict(perm=perm.astype(np.int64), L=L, z=z, fixed=nf.value, stats=dict(zip(keys, st)))


def symbolic_structures(A, wcap=32, lds=7680):
    """(perm, etree parent, entries per column of L) as csrc/sparse_symbolic.h computes them (amalgamation off), for
    the independent NumPy check of tests/test_sparse_symbolic.py."""
    A = sp.csc_matrix(A, dtype=np.float64)
    A.sort_indices()
    m, n = A.shape
    pi = C.POINTER(C.c_int32)
    perm, parent, cnt = (np.zeros(m, dtype=np.int32) for _ in range(3))
    ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
    ii = np.ascontiguousarray(A.indices, dtype=np.int32)
    rc = load().spsym_structures(m, n, ip.ctypes.data_as(pi), ii.ctypes.data_as(pi), wcap, lds, perm.ctypes.data_as(pi),
                                 parent.ctypes.data_as(pi), cnt.ctypes.data_as(pi))
    if rc:
        raise RuntimeError("spsym_structures failed with code %d" % rc)
    return perm.astype(np.int64), parent.astype(np.int64), cnt.astype(np.int64)
