"""CPU tests of the sparse-factor front end: ipm_order_rows (host-only C ABI) and the C++ oracle of the
multifrontal scheme over the same symbolic structures (oracle/sparse_chol_oracle.cpp), against dense LAPACK / SuperLU.
The reference gets this service from scipy's spsolve (SuperLU + COLAMD, main.py:180)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

from interiorpointmethod_amd import _lib
from interiorpointmethod_amd import solver as S
from interiorpointmethod_amd.matio import load_npz_problem

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _lp(name):
    A, b, c, cTlb, valid = load_npz_problem(os.path.join(G, "netlib", name + ".npz"))
    return sp.csc_matrix(A, dtype=np.float64), b, c


@pytest.fixture(scope="module")
def oracle(built_lib):
    from oracle import sparse_chol
    sparse_chol.load()
    return sparse_chol


def _nnz_chol(P, perm):
    """Entries of the Cholesky factor of the pattern P in the given order (symbolic, by child merging)."""
    Pp = sp.csc_matrix(P[perm][:, perm])
    m = Pp.shape[0]
    struct, kids, total = [None] * m, [[] for _ in range(m)], 0
    low = sp.tril(Pp, -1).tocsc()
    for k in range(m):
        s = set(low.indices[low.indptr[k]:low.indptr[k + 1]].tolist())
        for ch in kids[k]:
            s |= struct[ch]
            struct[ch] = None
        s.discard(k)
        struct[k] = s
        total += len(s) + 1
        if s:
            kids[min(s)].append(k)
    return total


@pytest.mark.parametrize("name", ["AFIRO", "SC205", "BANDM", "SCTAP1", "SHELL", "25FV47"])
def test_order_rows_is_a_permutation_with_the_fill_it_reports(built_lib, name):
    A, b, c = _lp(name)
    perm, info = S.sparse_factor_order(A)
    m = A.shape[0]
    assert sorted(perm.tolist()) == list(range(m))
    Ab = A.copy()
    Ab.data[:] = 1.0
    P = sp.csr_matrix(Ab @ Ab.T)
    P.data[:] = 1.0
    assert info["nnz_pattern"] == sp.tril(P, -1).nnz
    assert info["nnz_factor"] == _nnz_chol(P, perm)             # the reported fill is the fill of the returned order
    assert info["nnz_factor"] <= _nnz_chol(P, np.arange(m))     # ... and no worse than the natural order
    # same league as SuperLU's minimum-degree order on A A^T (the reference's spsolve uses COLAMD)
    from scipy.sparse.linalg import splu
    M = (P + sp.eye(m) * (m + 1.0)).tocsc()
    lu = splu(M, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
    assert info["nnz_factor"] <= 1.25 * lu.L.nnz
    perm2, info2 = S.sparse_factor_order(A)
    assert np.array_equal(perm, perm2) and info == info2          # a function of the pattern only


def test_order_rows_argument_checks(built_lib):
    lib = _lib.load()
    i32 = C.POINTER(C.c_int32)
    cp = np.array([0, 2, 3], dtype=np.int32)
    ri = np.array([0, 5, 1], dtype=np.int32)                       # row 5 out of range for m = 3
    perm = np.zeros(3, dtype=np.int32)
    rc = lib.ipm_order_rows(3, 2, cp.ctypes.data_as(i32), ri.ctypes.data_as(i32), perm.ctypes.data_as(i32), None)
    assert rc == -1 and b"out of range" in lib.ipm_last_error(None)
    ri[1] = 2
    rc = lib.ipm_order_rows(3, 2, cp.ctypes.data_as(i32), ri.ctypes.data_as(i32), perm.ctypes.data_as(i32), None)
    assert rc == 0 and sorted(perm.tolist()) == [0, 1, 2]


def test_order_rows_gives_up_on_a_dense_normal_matrix(built_lib):
    A, b, c = _lp("QAP8")                                          # A A^T fills to ~dense: the dense-tile path's business
    rng = np.random.default_rng(0)
    D = sp.csc_matrix(rng.standard_normal((2500, 40)))             # 2500 x 2500 dense pattern: over the ordering budget
    perm, info = S.sparse_factor_order(D)
    assert perm is None and info is None
    assert not S.prefer_sparse_factor(912, S.sparse_factor_order(A)[1], 8)


@pytest.mark.parametrize("name,wcap,lds", [("AFIRO", 32, 7680), ("SC205", 32, 7680), ("BANDM", 32, 7680), ("BANDM", 4, 64),
                                           ("SHARE2B", 3, 40), ("CZPROB", 32, 7680), ("SCTAP1", 8, 256)])
def test_multifrontal_oracle_against_lapack(oracle, name, wcap, lds):
    """L L^T = (A D A^T)(perm, perm) and B z = rhs, for the GPU's panel limits and for tiny ones that force panel splits."""
    A, b, c = _lp(name)
    rng = np.random.default_rng(3)
    d = rng.uniform(0.5, 2.0, A.shape[1])
    rhs = rng.standard_normal(A.shape[0])
    out = oracle.factor_solve(A, d, rhs, wcap=wcap, lds=lds)
    B = (A @ sp.diags(d) @ A.T).toarray()
    Bp = B[np.ix_(out["perm"], out["perm"])]
    assert out["fixed"] == 0
    assert np.abs(out["L"] @ out["L"].T - Bp).max() <= 1e-13 * np.abs(Bp).max()
    Lref = sla.cholesky(Bp, lower=True)
    assert np.abs(out["L"] - Lref).max() <= 1e-9 * np.abs(Lref).max()
    zref = np.linalg.solve(B, rhs)
    assert np.linalg.norm(out["z"] - zref) <= 1e-9 * np.linalg.norm(zref)
    assert out["stats"]["max_children"] <= 12                      # fan-in nodes bound the children of a panel


def test_multifrontal_oracle_fan_in_nodes_and_guard(oracle):
    A, b, c = _lp("CZPROB")                                        # a star: hundreds of leaves under one panel
    out = oracle.factor_solve(A, np.ones(A.shape[1]), np.ones(A.shape[0]), want_factor=False)
    assert out["stats"]["fan_in_nodes"] > 50 and out["stats"]["height"] <= 12
    # dependent rows: the guard replaces the pivot (LIPSOL rule, oracle/ipm_oracle.py::guarded_cholesky) and z stays finite
    A2 = sp.vstack([A[:40], A[:3]]).tocsc()
    out2 = oracle.factor_solve(A2, np.ones(A2.shape[1]), np.ones(A2.shape[0]), eps=1e-12)
    assert out2["fixed"] == 3 and np.all(np.isfinite(out2["z"]))
    from oracle import ipm_oracle as O
    B2 = (A2 @ A2.T).toarray()
    Lg, nfix = O.guarded_cholesky(B2[np.ix_(out2["perm"], out2["perm"])], eps=1e-12)
    assert nfix == 3
    good = np.abs(np.diag(Lg)) < 1e30
    assert good.sum() == 40 and np.allclose(np.diag(out2["L"])[good], np.diag(Lg)[good], rtol=1e-9)
    assert np.all(np.diag(out2["L"])[~good] == 1e32)


def test_factor_auto_rule_on_the_netlib_suite(built_lib):
    """Which LPs factor="auto" sends to the sparse path: the genuinely sparse large ones, not the dense-fill ones."""
    picks = {}
    for name in ("STOCFOR3", "SIERRA", "STOCFOR2", "CZPROB", "BNL2", "D2Q06C", "GREENBEA", "25FV47", "GROW22", "PILOT"):
        A, b, c = _lp(name)
        m = A.shape[0]
        perm, info = S.sparse_factor_order(A) if S._worth_ordering(A) else (None, None)
        picks[name] = S.prefer_sparse_factor(m, info, (m + 127) // 128)
    assert picks == dict(STOCFOR3=True, SIERRA=True, STOCFOR2=True, CZPROB=True, BNL2=False, D2Q06C=False, GREENBEA=False,
                         **{"25FV47": False}, GROW22=False, PILOT=False)


SWEEP = ["SCAGR7", "SC205", "SCSD6", "LOTFI", "FORPLAN", "E226", "BOEING2", "BORE3D", "SCTAP1", "SCFXM1", "SCORPION", "SCSD8", "GROW7",
         "DEGEN2", "SCAGR25", "STANDATA", "SCRS8", "FINNIS", "GFRD-PNC", "SHELL", "GROW15", "SCFXM3", "TRUSS", "SEBA",
         "SCTAP2", "WOODW", "CZPROB", "GROW22", "SCTAP3", "GANGES", "STOCFOR2", "NESM", "SIERRA", "80BAU3B", "STOCFOR3"]


@pytest.mark.parametrize("name", SWEEP)
def test_multifrontal_oracle_solves_the_normal_equations(oracle, name):
    """Every structure the symbolic layer produces over the Netlib set -- supernodes from 1 to 600 rows wide, relaxed
    amalgamation, panel splits, stars with hundreds of leaves (fan-in nodes), 16675 rows -- carries a correct solve:
    (A D A^T) z = rhs to 1e-6 in the residual, 1e-12 where A has full row rank (D strictly positive random; a shift of 1e-10 max diag keeps the handful of
    files with dependent rows away from the pivot guard, which test_multifrontal_oracle_fan_in_nodes_and_guard covers)."""
    A, b, c = _lp(name)
    rng = np.random.default_rng(11)
    d = rng.uniform(0.5, 2.0, A.shape[1])
    rhs = rng.standard_normal(A.shape[0])
    out = oracle.factor_solve(A, d, rhs, shift_rel=1e-10, want_factor=False)
    B = (A @ sp.diags(d) @ A.T).tocsr()
    shift = 1e-10 * B.diagonal().max()
    res = np.linalg.norm(B @ out["z"] + shift * out["z"] - rhs) / np.linalg.norm(rhs)
    assert out["fixed"] == 0 and res <= 1e-6, (name, out["fixed"], res, out["stats"])      # (a wrong index gives O(1))
    assert out["stats"]["max_children"] <= 12


def _numpy_etree_and_counts(P):
    """Independent restatement (shares no code with csrc/sparse_symbolic.h): Liu's elimination tree by ancestor path
    compression, and the entries of every column of L by explicit symbolic elimination (the structure of column k is its
    own below-diagonal pattern united with the structures of the columns whose first below-diagonal entry is k)."""
    P = sp.csc_matrix(P)
    m = P.shape[0]
    parent = -np.ones(m, dtype=np.int64)
    anc = -np.ones(m, dtype=np.int64)
    for i in range(m):
        for k in P.indices[P.indptr[i]:P.indptr[i + 1]]:
            while k != -1 and k < i:
                nxt = anc[k]
                anc[k] = i
                if nxt == -1:
                    parent[k] = i
                k = nxt
    below = [set(int(r) for r in P.indices[P.indptr[k]:P.indptr[k + 1]] if r > k) for k in range(m)]
    counts = np.zeros(m, dtype=np.int64)
    for k in range(m):
        counts[k] = len(below[k]) + 1
        if below[k]:
            p = min(below[k])
            below[p] |= below[k] - {p}
        below[k] = None
    return parent, counts


# the ten LPs the `factor="auto"` rule routes to the sparse factor (80BAU3B ... STOCFOR3: DESIGN 4-S), after four small ones
SPARSE_PATH_LPS = ["80BAU3B", "CZPROB", "GANGES", "GFRD-PNC", "SCTAP2", "SCTAP3", "SHELL", "SIERRA", "STOCFOR2", "STOCFOR3"]


@pytest.mark.parametrize("name", ["AFIRO", "SC205", "BANDM", "SCTAP1"] + SPARSE_PATH_LPS)
def test_etree_and_column_counts_against_an_independent_numpy_restatement(oracle, name):
    """The product's symbolic analysis (order -> elimination tree -> column structures by child merging -> panels) against
    a NumPy restatement that shares nothing with csrc/sparse_symbolic.h: same elimination-tree parent array, same number
    of entries in every column of L, the order is a postorder of that tree (parent after child, subtrees contiguous), and
    the first below-diagonal entry of a column is its parent."""
    A, b, c = _lp(name)
    perm, parent, cnt = oracle.symbolic_structures(A)
    m = A.shape[0]
    assert sorted(perm.tolist()) == list(range(m))
    Ab = A.copy()
    Ab.data[:] = 1.0
    P = sp.csr_matrix(Ab @ Ab.T)[perm][:, perm]
    ref_parent, ref_cnt = _numpy_etree_and_counts(P)
    assert np.array_equal(parent, ref_parent)
    assert np.array_equal(cnt, ref_cnt)
    assert int(cnt.sum()) == S.sparse_factor_order(A)[1]["nnz_factor"]
    # postorder: every vertex precedes its parent and each subtree is one contiguous index range ending at its root
    size = np.ones(m, dtype=np.int64)
    for k in range(m):
        if parent[k] >= 0:
            assert parent[k] > k
            size[parent[k]] += size[k]
    first = np.arange(m) - size + 1
    for k in range(m):
        if parent[k] >= 0:
            assert first[parent[k]] <= first[k]


ASAN_SET = ["AFIRO", "SC105", "SC205", "BANDM", "E226", "SCTAP1", "SHELL", "25FV47", "CZPROB", "STOCFOR2"]


def test_sparse_symbolic_and_oracle_under_address_and_ub_sanitizers(tmp_path):
    """csrc/sparse_symbolic.h (host code of the PRODUCT: ordering, elimination tree, amalgamation, panels, fan-in nodes)
    together with the C++ oracle, compiled with -fsanitize=address,undefined on the CPU (GPU sanitizers are not
    available on the pool) and run over ten Netlib structures at two panel budgets (the GPU's 32 x 7680 and a tiny one
    that forces panel splits and fan-in nodes): no report, exit status 0."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(tmp_path, "asan_sparse_driver")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(root, "interiorpointmethod_amd", "csrc"), "-I" + os.path.join(root, "oracle"),
           os.path.join(root, "tests", "asan_sparse_driver.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    for name in ASAN_SET:
        A, b, c = _lp(name)
        A.sort_indices()
        f = os.path.join(tmp_path, name + ".bin")
        with open(f, "wb") as fh:
            np.array([A.shape[0], A.shape[1], A.nnz], dtype=np.int32).tofile(fh)
            A.indptr.astype(np.int32).tofile(fh)
            A.indices.astype(np.int32).tofile(fh)
            A.data.astype(np.float64).tofile(fh)
        out = subprocess.run([exe, f], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
        assert out.returncode == 0, (name, out.stdout[-2000:], out.stderr[-4000:])
        assert "ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
