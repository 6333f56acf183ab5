"""General-form front end (interiorpointmethod_amd/general_form.py) against the reference's own conversion
(fixtures from tests/golden/make_golden_general.py) and, for the branches the reference leaves unfinished,
against scipy's HiGHS on the converted problem.  CPU only."""
import glob
import os

import numpy as np
import pytest
from scipy import sparse
from scipy.optimize import linprog

from interiorpointmethod_amd import general_form as G


def load_fixture(path):
    z = np.load(path)

    def mat(prefix):
        if prefix + "_none" in z.files:
            return None
        return sparse.csc_matrix((z[prefix + "_data"], z[prefix + "_indices"], z[prefix + "_indptr"]),
                                 shape=tuple(int(v) for v in z[prefix + "_shape"]))
    inp = dict(c=z["c"], Aineq=mat("Aineq"), bineq=z["bineq"] if "bineq" in z.files else None,
               Aeq=mat("Aeq"), beq=z["beq"] if "beq" in z.files else None, lb=z["lb"], ub=z["ub"])
    return z, inp, mat


def same_matrix(A, B):
    A, B = sparse.csc_matrix(A), sparse.csc_matrix(B)
    return A.shape == B.shape and (A != B).nnz == 0


FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "general", "*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_conversion_matches_reference(path):
    """get_Abc(options="no-bound") and add_bound_into_matrix reproduce the reference's matrices exactly."""
    z, inp, mat = load_fixture(path)
    A, b, c, bound = G.get_Abc(options="no-bound", **inp)
    assert same_matrix(A, mat("std0_A")) and np.array_equal(b, z["std0_b"]) and np.array_equal(c, z["std0_c"])
    assert (bound is None) == bool(z["bound_is_none"])
    if bound is not None:
        assert bound[0] is None                                   # "no-bound" drops finite lower bounds (main.py:896)
        A, b, c, rest, const = G.add_bound_into_matrix(A, b, c, bound)
        assert rest == (None, None) and float(np.asarray(const).ravel()[0] if np.ndim(const) else const) == float(z["const"])
    assert same_matrix(A, mat("std_A")) and np.array_equal(b, z["std_b"]) and np.array_equal(c, z["std_c"])
    A2, b2, c2, offset = G.standard_form(**inp)                   # what new_interior_sparse() hands to the solver
    if np.count_nonzero(z["lb"]) == 0:
        assert same_matrix(A2, A) and np.array_equal(b2, b) and np.array_equal(c2, c) and offset == 0.0
    else:       # nonzero lower bounds: the reference drops them here (main.py:896-899); the front end shifts instead
        assert A2.shape[1] == A.shape[1] and offset == pytest.approx((z["c"].T @ z["lb"]).item())


def test_get_Abc_shapes_and_options():
    c = np.array([-300.0, -500.0, -200.0])
    Aineq = np.array([[10, 7.5, 4], [0, 10, 0], [0.5, 0.4, 0.5]])
    bineq = np.array([4350.0, 2500.0, 280.0])
    lb, ub = np.zeros(3), np.array([300.0, 180.0, np.inf])
    for Ai in (Aineq, sparse.csc_matrix(Aineq)):
        A, b, cs, bound = G.get_Abc(c, Aineq=Ai, bineq=bineq, lb=lb, ub=ub, options="bound")
        assert sparse.issparse(A) == sparse.issparse(Ai) and A.shape == (3, 6) and b.shape == (3, 1) and cs.shape == (6, 1)
        dense = A.toarray() if sparse.issparse(A) else A
        assert np.array_equal(dense, np.hstack([Aineq, np.eye(3)])) and np.array_equal(cs[3:], np.zeros((3, 1)))
        assert bound[0] is None and np.array_equal(bound[1].ravel(), [300, 180, np.inf, np.inf, np.inf, np.inf])
        assert len(G.get_Abc(c, Aineq=Ai, bineq=bineq, lb=lb, ub=ub, options="no-bound")) == 4   # the reference returns 3
    Aeq, beq = np.array([[1.0, 1.0, 1.0]]), np.array([400.0])
    A, b, cs, bound = G.get_Abc(c, Aeq=Aeq, beq=beq, Aineq=Aineq, bineq=bineq, lb=lb, ub=np.full(3, np.inf))
    assert bound is None and A.shape == (4, 6) and np.array_equal(A[3], [1, 1, 1, 0, 0, 0]) and np.array_equal(b.ravel(), [4350, 2500, 280, 400])
    A, b, cs, bound = G.get_Abc(c, Aeq=Aeq, beq=beq, lb=np.array([1.0, 0, 0]), ub=np.full(3, np.inf), options="bound")
    assert A is Aeq and bound[1] is None and np.array_equal(bound[0].ravel(), [1, 0, 0])
    with pytest.raises(ValueError):
        G.get_Abc(c, Aeq=Aeq, beq=beq, lb=np.array([-np.inf, 0, 0]), ub=np.full(3, np.inf), options="no-bound")
    with pytest.raises(ValueError):
        G.get_Abc(c, lb=lb, ub=ub)


@pytest.mark.parametrize("case", ["upper", "lower", "both"])
def test_bounds_folded_correctly(case):
    """The standard-form problem has the same optimum as the bounded one (HiGHS on both); exercises the lower-bound
    shift (b - A lb; the reference adds, main.py:1047) and the two-sided branch the reference never finished."""
    rng = np.random.default_rng(4)
    n, mi, me = 9, 5, 2
    Aineq, Aeq = rng.uniform(0, 1, (mi, n)), rng.uniform(0, 1, (me, n))
    x0 = rng.uniform(1.0, 2.0, n)
    bineq, beq = Aineq @ x0 + 0.5, Aeq @ x0
    c = rng.standard_normal(n)
    lb = np.zeros(n) if case == "upper" else rng.uniform(0.2, 0.9, n)
    ub = np.full(n, np.inf) if case == "lower" else rng.uniform(2.1, 3.0, n)
    if case != "lower":
        ub[::3] = np.inf
    want = linprog(c, A_ub=Aineq, b_ub=bineq, A_eq=Aeq, b_eq=beq, bounds=list(zip(lb, ub)), method="highs")
    assert want.status == 0
    A, b, cs, offset = G.standard_form(c, Aeq=sparse.csc_matrix(Aeq), beq=beq, Aineq=sparse.csc_matrix(Aineq), bineq=bineq, lb=lb, ub=ub)
    got = linprog(cs.ravel(), A_eq=A, b_eq=b.ravel(), bounds=[(0, None)] * A.shape[1], method="highs")
    assert got.status == 0 and abs(got.fun + offset - want.fun) <= 1e-8 * (1 + abs(want.fun))
    assert A.shape == (mi + me + int(np.isfinite(ub).sum()), n + mi + int(np.isfinite(ub).sum()))
