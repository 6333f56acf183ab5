"""GPU tests (-m gpu) that DRIVE the recovery paths of the device-side hand-offs on purpose.

Every cross-workgroup / cross-stream hand-off of the library is a bounded spin on a device counter; a wait that runs into its
bound sets the handle's time-out word, the host rolls the call back to its snapshot and repeats it on a path that does not poll,
and the handle keeps that path (csrc/ipm_api.hip: read_scalars / poll_fallback).  No committed run had ever taken that path.
The test knob IPM_TEST_SPIN_LIMIT (read at ipm_create, csrc/gemm_nt_f64.h) lowers the bound to a few polls, so that the FIRST
wait of each kind that really has to wait gives up.  One run per path: (a) the fused formation + factorization launch, (b) the
polled look-ahead of the blocked Cholesky, (c) the task hand-offs of the sparse multifrontal factor.  Checked: the call completes,
the recovery is counted, the handle stays on the non-polling path, and the result is BIT-IDENTICAL to a handle that was put on
that path from the start (fp64; the paths share every kernel and summation order).  Reference semantics: main.py:224-226."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import interiorpointmethod_amd as ipm                              # noqa: E402
from interiorpointmethod_amd import _lib                           # noqa: E402
from interiorpointmethod_amd.matio import load_npz_problem         # noqa: E402
from interiorpointmethod_amd.workloads import synthetic_lp         # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run_dense(A, b, c, steps, **kw):
    with ipm.IpmSolver(A, b, c, **kw) as sv:
        sv.init_state(0.0)
        st = sv.iterate(steps)
        x, y, s = sv.get_state()
        sch = sv.schedule()
    return st, x, y, s, sch


def test_fused_launch_timeout_is_recovered(monkeypatch):
    """(a) 16 blocks: fused by default.  With a spin bound of 8 polls the persistent launch gives up, the call is repeated with
    formation + look-ahead factorization (counters), which gives up too, and then with stream events: two recoveries, fused off."""
    A, b, c = synthetic_lp(2048, 4100, seed=3)
    monkeypatch.setenv("IPM_FUSED_FACTOR", "0")
    monkeypatch.setenv("IPM_FLAG_SYNC", "0")                       # reference: stream events from the start
    st0, x0, y0, s0, sch0 = _run_dense(A, b, c, 3)
    assert sch0["fused_factor"] == 0 and sch0["device_polling"] == 0 and sch0["timeouts_recovered"] == 0
    monkeypatch.delenv("IPM_FUSED_FACTOR"); monkeypatch.delenv("IPM_FLAG_SYNC")
    monkeypatch.setenv("IPM_TEST_SPIN_LIMIT", "8")
    st1, x1, y1, s1, sch1 = _run_dense(A, b, c, 3)
    monkeypatch.delenv("IPM_TEST_SPIN_LIMIT")
    assert sch1["timeouts_recovered"] == 2, sch1
    assert sch1["fused_factor"] == 0 and sch1["device_polling"] == 0, sch1      # the handle stays on the non-polling path
    assert st1["iterations"] == 3 and np.isfinite(st1["objective"])
    assert np.array_equal(x1, x0) and np.array_equal(y1, y0) and np.array_equal(s1, s0)
    assert st1["objective"] == st0["objective"]
    # a fresh handle afterwards has the default bound again and runs fused without a time-out
    st2, x2, y2, s2, sch2 = _run_dense(A, b, c, 3)
    assert sch2["fused_factor"] == 1 and sch2["timeouts_recovered"] == 0


def test_lookahead_poll_timeout_is_recovered(monkeypatch):
    """(b) the polled look-ahead of the blocked Cholesky (fused launch off): one recovery, stream events from then on."""
    A, b, c = synthetic_lp(1500, 3100, seed=4)
    monkeypatch.setenv("IPM_FUSED_FACTOR", "0")
    monkeypatch.setenv("IPM_FLAG_SYNC", "0")
    st0, x0, y0, s0, sch0 = _run_dense(A, b, c, 4)
    monkeypatch.delenv("IPM_FLAG_SYNC")
    monkeypatch.setenv("IPM_TEST_SPIN_LIMIT", "4")
    st1, x1, y1, s1, sch1 = _run_dense(A, b, c, 4)
    monkeypatch.delenv("IPM_TEST_SPIN_LIMIT")
    assert sch1["timeouts_recovered"] == 1 and sch1["device_polling"] == 0, sch1
    assert np.array_equal(x1, x0) and np.array_equal(y1, y0) and np.array_equal(s1, s0)
    assert st1["objective"] == st0["objective"] and st1["iterations"] == 4
    # the whole solve through the solver seam with the tiny bound: converges to the same answer as without it
    monkeypatch.setenv("IPM_TEST_SPIN_LIMIT", "4")
    xa, ya, sa, ia = ipm.solve_with_info(A, b, c, tol=1e-8, y0=0.0, max_iter=100)
    monkeypatch.delenv("IPM_TEST_SPIN_LIMIT")
    xb, yb, sb, ib = ipm.solve_with_info(A, b, c, tol=1e-8, y0=0.0, max_iter=100)
    assert ia["status"] == 1 and ia["iterations"] == ib["iterations"]
    assert abs(ia["objective"] - ib["objective"]) <= 1e-9 * max(1.0, abs(ib["objective"]))


def test_sparse_sweep_timeout_is_recovered(monkeypatch):
    """(c) the task hand-offs inside the sparse factor's sweeps (one launch per sweep, flags between tasks): after the time-out the
    handle launches ONE workgroup per sweep, which never waits (ipm_get_factor_info: serial_launches > 0); same bits as the
    undisturbed run (the sweeps are schedule independent, tests/test_gpu_sparse_factor.py)."""
    A, b, c, _, valid = load_npz_problem(os.path.join(GOLDEN, "netlib", "SCTAP2.npz"))
    assert valid
    monkeypatch.setenv("IPM_SP_MODE", "task")                        # the polling form (a lone handle picks it anyway)

    def run():
        with ipm.IpmSolver(A, b, c, factor="sparse") as sv:
            assert sv.factor == "sparse"
            sv.init_state(1.0)
            st = sv.iterate(5)
            x, y, s = sv.get_state()
            return st, x, y, s, sv.schedule(), sv.factor_info()
    st0, x0, y0, s0, sch0, fi0 = run()
    assert sch0["timeouts_recovered"] == 0 and fi0["serial_launches"] == 0
    monkeypatch.setenv("IPM_TEST_SPIN_LIMIT", "2")
    st1, x1, y1, s1, sch1, fi1 = run()
    monkeypatch.delenv("IPM_TEST_SPIN_LIMIT")
    assert sch1["timeouts_recovered"] == 1, sch1
    assert fi1["serial_launches"] > 0
    assert np.array_equal(x1, x0) and np.array_equal(y1, y0) and np.array_equal(s1, s0)
    assert st1["objective"] == st0["objective"]
