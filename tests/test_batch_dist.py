"""Batched-LP mode on CPU: LPT partition, and the N>1 statistics gather over gloo (world_size 2).
The per-LP solve is stubbed (no GPU here); what is under test is the sharding and the single
collective of interiorpointmethod_amd/batch.py."""
import os
import socket

import numpy as np
import pytest

from interiorpointmethod_amd import batch


def test_lpt_partition_balanced_and_deterministic():
    costs = [1.55e12, 8.4e10, 7.5e10, 6.3e10, 1.9e10, 1.2e10, 1.1e10, 1.0e10, 7e9, 6e9, 5e9, 4e9]
    for world in (1, 2, 4, 8):
        shards = batch.lpt_partition(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(len(costs)))
        assert shards == batch.lpt_partition(list(costs), world)
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) == pytest.approx(max(costs[0], sum(costs) / world), rel=0.35)
    # the dominant problem sits alone when ranks allow (STOCFOR3 bounds the makespan)
    assert batch.lpt_partition(costs, 8)[0] == [0]


def test_predicted_cost_orders_stocfor3_first():
    assert batch.predicted_cost(16675, 23541, 4.5e5) > batch.predicted_cost(6330, 22275, 5.6e5)


def _fake_solve(problem, device=0, **kw):
    m, n = problem[0].shape
    return dict(status=1 if m % 2 == 0 else 2, iterations=m + n, objective=float(m) / n, rp=1e-9, rd=2e-9,
                gap=3e-9, pivots_fixed=m % 3)


def test_single_process_batch():
    probs = [(np.zeros((m, m + 1)), None, None) for m in (4, 9, 2, 7, 6)]
    rec, secs = batch.run_batch(probs, solve_fn=_fake_solve)
    assert rec.shape == (5, batch.NF) and list(rec[:, 0]) == [0, 1, 2, 3, 4]
    s = batch.summarize(rec)
    assert s["n"] == 5 and s["converged"] == 3 and s["max_iter"] == 2
    assert s["total_iterations"] == sum(2 * m + 1 for m in (4, 9, 2, 7, 6))


def _raising_solve(problem, device=0, **kw):
    """One LP of the batch fails with something that is NOT a library error (bad shapes, OOM, ...)."""
    if problem[0].shape[0] == 7:
        raise ValueError("b has length 3, expected 7")
    return _fake_solve(problem, device=device, **kw)


def _worker(rank, world, port, q, schedule="static", workers=1, solve_fn=_fake_solve):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # the counter of the self-scheduling mode lives on an explicit TCPStore (public API), shared with the group
    store = batch.make_store(rank, world, host="127.0.0.1", port=port)
    dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
    try:
        probs = [(np.zeros((m, m + 1)), None, None) for m in (4, 9, 2, 7, 6, 11, 3)]
        rec, secs = batch.run_batch(probs, dist=dist, solve_fn=solve_fn, schedule=schedule, workers=workers, store=store)
        rec2, _ = batch.run_batch(probs, dist=dist, solve_fn=solve_fn, schedule=schedule, workers=workers, store=store)   # a second call: fresh counter
        assert np.array_equal(rec[:, :3], rec2[:, :3])
        q.put((rank, rec))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("schedule,workers", [("static", 1), ("dynamic", 2)])
def test_two_rank_gloo_one_lp_raises(schedule, workers):
    """An exception inside one solve must not keep its rank from the all-gather (the peers would block in the
    collective until the process-group time-out): it becomes a STATUS_ERROR record, every other LP is solved."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, schedule, workers, _raising_solve)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        rec = got[r]
        assert list(rec[:, 0]) == list(range(7))
        assert rec[3, 1] == batch.STATUS_ERROR and np.isnan(rec[3, 3])
        assert np.all(rec[[0, 1, 2, 4, 5, 6], 1] > 0)
    assert batch.summarize(got[0])["errors"] == 1


@pytest.mark.parametrize("schedule,workers", [("dynamic", 1), ("dynamic", 2)])
def test_two_rank_gloo_dynamic_schedule(schedule, workers):
    """Self-scheduling from the rendezvous store: every LP solved exactly once, same table on both ranks."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, schedule, workers)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, _ = batch.run_batch([(np.zeros((m, m + 1)), None, None) for m in (4, 9, 2, 7, 6, 11, 3)], solve_fn=_fake_solve)
    for r in (0, 1):
        assert np.array_equal(got[r][:, :7], ref[:, :7]) and np.array_equal(got[r][:, 8], ref[:, 8])


def test_two_rank_gloo_gather():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, _ = batch.run_batch([(np.zeros((m, m + 1)), None, None) for m in (4, 9, 2, 7, 6, 11, 3)],
                             solve_fn=_fake_solve)
    ref[:, 7] = 0
    for r in (0, 1):                      # every rank holds the full, id-ordered table
        rec = got[r].copy()
        rec[:, 7] = 0                     # wall seconds differ
        assert np.array_equal(rec, ref)
