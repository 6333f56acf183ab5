"""GPU tests (-m gpu) of the LOCKSTEP BATCH (csrc/lockstep.h, ipm_solve_batch): iteration k of several independent LPs in the same
launches -- the driver loop of the reference (script.py:147-173 over the Netlib files) as one chain of launches instead of one
chain per LP.  A handle's arithmetic in the batch is the body of the very kernels it would launch alone, with the same arguments
in the same order, so every LP of a batch must end BIT-IDENTICAL to the same handle solved alone with ipm_solve: status,
iteration count, objective, residuals, (x, y, s).  fp64; bit-exact asserts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import interiorpointmethod_amd as ipm                              # noqa: E402
from interiorpointmethod_amd.matio import load_npz_problem         # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "netlib")


def _load(nm):
    A, b, c, _, valid = load_npz_problem(os.path.join(GOLDEN, nm + ".npz"))
    assert valid
    return A, b, c


def _alone(A, b, c, max_iter):
    with ipm.IpmSolver(A, b, c, lockstep=True, factor="dense") as sv:
        sv.init_state(1.0)
        st = sv.solve(tol=1e-8, max_iter=max_iter)
        return st, sv.get_state()


@pytest.mark.parametrize("names,max_iter", [
    (["BANDM", "SCFXM1", "E226", "SC205"], 300),                              # 2-3 blocks, product-list formation
    (["DEGEN3", "BNL1", "WOODW", "25FV47", "SEBA", "BANDM"], 120),           # 3 ... 12 blocks, with and without a tile envelope
    (["BNL2", "MAROS-R7", "TRUSS", "SCSD8", "D2Q06C"], 60),                  # envelope LPs, rows above 1536 (row-owner formation)
])
def test_lockstep_batch_is_bit_identical_to_one_at_a_time(names, max_iter):
    probs = [_load(nm) for nm in names]
    ref = [_alone(*p, max_iter) for p in probs]
    svs = [ipm.IpmSolver(*p, lockstep=True, factor="dense") for p in probs]
    try:
        for sv in svs:
            assert ipm.lockstep_eligible(sv)
            sv.init_state(1.0)
        stats = ipm.solve_lockstep(svs, tol=1e-8, max_iter=max_iter)
        for nm, sv, st, (st0, (x0, y0, s0)) in zip(names, svs, stats, ref):
            x, y, s = sv.get_state()
            assert st["status"] == st0["status"] and st["iterations"] == st0["iterations"], (nm, st, st0)
            for k in ("objective", "rp_norm", "rd_norm", "gap", "pivots_fixed", "auto_regularized"):
                assert st[k] == st0[k] or (np.isnan(st[k]) and np.isnan(st0[k])), (nm, k, st[k], st0[k])
            assert np.array_equal(x, x0, equal_nan=True) and np.array_equal(y, y0, equal_nan=True) and np.array_equal(s, s0, equal_nan=True), nm
    finally:
        for sv in svs:
            sv.close()


def test_lockstep_batch_restarts_a_rank_deficient_lp_with_the_automatic_shift():
    """QAP8 has 13 % dependent rows: ipm_solve switches the 1e-14 Tikhonov shift on after the first factorization and restarts; in
    a batch that restart must touch only that LP (its program is re-recorded), the neighbours run on."""
    names = ["QAP8", "BANDM", "SCFXM1"]
    probs = [_load(nm) for nm in names]
    ref = [_alone(*p, 100) for p in probs]
    assert ref[0][0]["auto_regularized"] == 1 and ref[0][0]["status"] == 1
    svs = [ipm.IpmSolver(*p, lockstep=True, factor="dense") for p in probs]
    try:
        for sv in svs:
            sv.init_state(1.0)
        stats = ipm.solve_lockstep(svs, tol=1e-8, max_iter=100)
        for nm, sv, st, (st0, (x0, y0, s0)) in zip(names, svs, stats, ref):
            x, y, s = sv.get_state()
            assert (st["status"], st["iterations"], st["auto_regularized"]) == (st0["status"], st0["iterations"], st0["auto_regularized"]), nm
            assert st["objective"] == st0["objective"] and np.array_equal(x, x0) and np.array_equal(y, y0), nm
    finally:
        for sv in svs:
            sv.close()


def test_lockstep_rejects_handles_it_cannot_serve():
    A, b, c = _load("AFIRO")                                   # 27 rows: the fused single-workgroup kernel serves it
    with ipm.IpmSolver(A, b, c, lockstep=True) as sv:
        sv.init_state(1.0)
        assert not ipm.lockstep_eligible(sv)
        with pytest.raises(ipm.IpmError):
            ipm.solve_lockstep([sv])
    A, b, c = _load("BANDM")
    with ipm.IpmSolver(A, b, c) as sv:                          # not created for the batch
        sv.init_state(1.0)
        with pytest.raises(ipm.IpmError):
            ipm.solve_lockstep([sv])


def test_lockstep_batch_lets_an_lp_join_between_two_steps():
    """ipm_batch_add between two ipm_batch_step calls: the late LP starts at ITS iteration 0 next to neighbours that are several
    iterations in, and still ends bit-identical to the handle solved alone (batch.solve_shard_lockstep feeds its batches this way
    while the set-up threads are still preparing the next LPs)."""
    names = ["SCFXM1", "BANDM", "DEGEN3", "E226"]
    probs = [_load(nm) for nm in names]
    ref = [_alone(*p, 300) for p in probs]
    svs = [ipm.IpmSolver(*p, lockstep=True, factor="dense") for p in probs]
    try:
        for sv in svs:
            sv.init_state(1.0)
        done = []
        with ipm.LockstepBatch(tol=1e-8, max_iter=300) as bt:
            bt.add(svs[0]); bt.add(svs[1])
            done += bt.step(); done += bt.step()
            bt.add(svs[2])
            done += bt.step()
            bt.add(svs[3])
            for _ in range(400):
                if not bt.active:
                    break
                done += bt.step()
            assert bt.active == 0
        assert sorted(id(s) for s in done) == sorted(id(s) for s in svs)
        for nm, sv, (st0, (x0, y0, s0)) in zip(names, svs, ref):
            x, y, s = sv.get_state()
            st = sv.stats
            assert (st["status"], st["iterations"]) == (st0["status"], st0["iterations"]), (nm, st, st0)
            assert st["objective"] == st0["objective"] or (np.isnan(st["objective"]) and np.isnan(st0["objective"])), nm
            assert np.array_equal(x, x0, equal_nan=True) and np.array_equal(y, y0, equal_nan=True) and np.array_equal(s, s0, equal_nan=True), nm
    finally:
        for sv in svs:
            sv.close()


def test_run_batch_lockstep_equals_the_one_at_a_time_table():
    """batch.run_batch(lockstep=True) -- what bench.py's Netlib legs run: one batch per size class, the small and the large
    sparse-factor LPs on their own kernels beside them -- must print the table of the one-at-a-time driver loop
    (script.py:147-173).  An LP that went through a batch: status, iteration count and objective BIT-IDENTICAL to the same
    lockstep handle (dense-tile factor) solved alone; an LP outside the batches (up to 128 rows): the row of run_batch(workers=1)."""
    from interiorpointmethod_amd import batch
    names = ["AFIRO", "ADLITTLE", "BANDM", "SCFXM1", "E226", "DEGEN3", "BNL1", "WOODW", "25FV47", "SEBA", "SC205", "GFRD-PNC",
             "SCTAP3", "SHELL", "QAP8", "TRUSS", "SCSD8", "STOCFOR2", "SC50A", "KB2"]
    probs = [_load(nm) for nm in names]
    ls, _ = batch.run_batch(probs, tol=1e-8, max_iter=300, workers=8, lockstep=True)
    F = batch.RECORD_FIELDS
    assert np.array_equal(ls[:, 0], np.arange(len(names))) and set(ls[:, 1].tolist()) <= {1.0, 2.0, 3.0}
    assert not ls[:, F.index("timeouts_recovered")].any() and not ls[:, F.index("serial_launches")].any()
    assert np.all(ls[:, F.index("solve_seconds")] > 0)
    small = [k for k, p in enumerate(probs) if p[0].shape[0] <= 128]
    assert len(small) >= 4
    seq, _ = batch.run_batch([probs[k] for k in small], tol=1e-8, max_iter=300, workers=1)
    for k, a in zip(small, seq):
        b = ls[k]
        assert np.array_equal(a[1:3], b[1:3]) and (a[3] == b[3] or (np.isnan(a[3]) and np.isnan(b[3]))), (names[k], a[1:4], b[1:4])
    for k, p in enumerate(probs):
        if k in small:
            continue
        st, _ = _alone(*p, 300)
        b = ls[k]
        assert (st["status"], st["iterations"]) == (int(b[1]), int(b[2])), (names[k], st, b[1:4])
        assert st["objective"] == b[3] or (np.isnan(st["objective"]) and np.isnan(b[3])), (names[k], st["objective"], b[3])
