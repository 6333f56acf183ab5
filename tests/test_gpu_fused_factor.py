"""GPU tests (-m gpu) of the FUSED formation + factorization (csrc/form_factor.h, csrc/ff_schedule.h): one persistent
worker launch beside the pivot chain instead of formation followed by factorization (main.py:224 + the factorization inside
main.py:180 / :226 of the reference).  The serial path (IPM_FUSED_FACTOR=0), itself pinned against the reference's golden
vectors by tests/test_gpu_parity.py, is the checker here; IPM_FUSED_FACTOR=force runs the fused kernels below their
default size limit of 16 blocks so that small and ragged sizes are covered too.  fp64; bounds next to each assert."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import interiorpointmethod_amd as ipm                              # noqa: E402
from interiorpointmethod_amd.workloads import synthetic_lp         # noqa: E402


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, float(np.max(np.abs(b)))))


def _run(A, b, c, steps):
    with ipm.IpmSolver(A, b, c) as sv:
        sv.init_state(0.0)
        st = sv.iterate(steps)
        x, y, s = sv.get_state()
        L = sv.get_factor()
        sched = sv.schedule()
    return st, x, y, s, L, sched


@pytest.mark.parametrize("m,n", [(400, 900), (512, 1100), (1100, 2300), (2048, 4100), (2500, 5000)])
def test_fused_path_matches_the_serial_path(monkeypatch, m, n):
    """Three iterations from the start point: same iterate (1e-9 relative), same Cholesky factor of the last normal
    matrix (1e-10), same scalars; the fused run repeats bit for bit; no hand-off timed out."""
    A, b, c = synthetic_lp(m, n, seed=5)
    monkeypatch.setenv("IPM_FUSED_FACTOR", "0")
    st0, x0, y0, s0, L0, sch0 = _run(A, b, c, 3)
    assert sch0["fused_factor"] == 0
    monkeypatch.setenv("IPM_FUSED_FACTOR", "force")
    st1, x1, y1, s1, L1, sch1 = _run(A, b, c, 3)
    assert sch1["fused_factor"] == 1 and sch1["timeouts_recovered"] == 0
    assert rel(L1, L0) < 1e-10
    assert rel(x1, x0) < 1e-9 and rel(y1, y0) < 1e-9 and rel(s1, s0) < 1e-9
    assert abs(st1["objective"] - st0["objective"]) <= 1e-10 * max(1.0, abs(st0["objective"]))
    assert abs(st1["alpha_p"] - st0["alpha_p"]) <= 1e-9 and abs(st1["sigma"] - st0["sigma"]) <= 1e-9
    st2, x2, y2, s2, L2, _ = _run(A, b, c, 3)
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2) and np.array_equal(L1, L2)


def test_fused_path_full_solve_and_factor_against_lapack(monkeypatch):
    """A whole solve on the fused path at 21 blocks (default rule: on from 20 blocks, without `force`): converges like the serial path
    (same iteration count, objective 1e-9), the LP-level properties hold on the host, and the factor the handle holds
    after convergence is the complete Cholesky factor of the final normal matrix (L L^T = A D^2 A^T to 1e-12: the latch
    of the overlapped path keeps the last factorization whole)."""
    m, n = 2600, 5300
    A, b, c = synthetic_lp(m, n, seed=11)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("IPM_FUSED_FACTOR", mode)
        with ipm.IpmSolver(A, b, c) as sv:
            sv.init_state(0.0)
            st = sv.solve(tol=1e-8, max_iter=200)
            x, y, s = sv.get_state()
            res[mode] = (st, x, y, s, sv.schedule())
    st0, st1 = res["0"][0], res["1"][0]
    assert res["1"][4]["fused_factor"] == 1 and res["0"][4]["fused_factor"] == 0 and res["1"][4]["timeouts_recovered"] == 0
    assert st0["status"] == 1 and st1["status"] == 1 and st1["iterations"] == st0["iterations"]
    assert abs(st1["objective"] - st0["objective"]) <= 1e-9 * max(1.0, abs(st0["objective"]))
    _, x, y, s, _ = res["1"]
    assert np.linalg.norm(A @ x - b) / (1 + np.linalg.norm(b)) < 1e-8
    assert np.linalg.norm(A.T @ y + s - c) / (1 + np.linalg.norm(c)) < 1e-8
    assert np.all(x > 0) and np.all(s > 0) and (x.T @ s).item() <= 1e-8


def test_fused_path_factor_of_a_given_scaling(monkeypatch):
    """One iteration from a chosen (x, s): the factor left in the handle is the Cholesky factor of A diag(x/s) A^T
    (LAPACK on the host, 1e-10), i.e. formation chunks, updates, panel solves and the chain's own kernels produce one
    consistent factor."""
    m, n = 2048, 4096
    rng = np.random.default_rng(3)
    A, b, c = synthetic_lp(m, n, seed=2)
    x = rng.uniform(0.5, 2.0, n); s = rng.uniform(0.5, 2.0, n); y = rng.standard_normal(m)
    monkeypatch.setenv("IPM_FUSED_FACTOR", "force")
    with ipm.IpmSolver(A, b, c) as sv:
        sv.set_state(x, y, s)
        sv.iterate(1)
        L = sv.get_factor()
        assert sv.schedule()["fused_factor"] == 1
    B = (A * (x / s)) @ A.T
    Lref = np.linalg.cholesky(B)
    assert rel(L, Lref) < 1e-10


def test_default_rule_selects_the_fused_path_where_it_was_measured_faster():
    """The selection rule (ipm_api.hip: 16 .. 72 blocks, n <= 6 m; profiles/r04_ff_sizes_fused_vs_serial.txt): the headline size and
    a 16-block LP run fused, a very wide LP (formation-dominated) and a small one do not."""
    for (m, n), want in (((4096, 8192), 1), ((2048, 4096), 1), ((1536, 3072), 0), ((2048, 16384), 0)):
        A, b, c = synthetic_lp(m, n, seed=1)
        with ipm.IpmSolver(A, b, c) as sv:
            sv.init_state(0.0)
            sv.iterate(1)
            assert sv.schedule()["fused_factor"] == want, (m, n)
