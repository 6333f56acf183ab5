"""Batched-LP mode: independent LP instances sharded across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for
tests).  An LP is solved entirely on one GPU -- there is no exchange inside a solve -- so the
only collective is ONE all-gather of a fixed-size statistics record per LP at the end
(SURVEY.md 8e).  The partition is either a deterministic LPT-greedy assignment on a predicted cost,
computed identically on every rank (no scheduling traffic at all), or self-scheduling from an atomic counter on
the process group's rendezvous store (control plane; balances the unpredictable iteration counts).

The reference has no counterpart (it is single process); the loop being distributed is the
driver loop of script.py:147-173 over the Netlib files.
"""
from __future__ import annotations

import time

import numpy as np

# One record per LP.  The first nine fields are the statistics SURVEY.md 8e names; the rest make the library's hidden
# recoveries and the host-side phases of a solve visible in the gathered table: `timeouts_recovered` = device hand-off
# polls that gave up and were rolled back and repeated (ipm_get_schedule), `serial_launches` = sparse-factor sweeps
# that ran as ONE workgroup after such a time-out (ipm_get_factor_info), setup / solve / teardown = host seconds of
# handle creation + upload + symbolic analysis, of ipm_solve, and of read-back + destroy (solver.solve_with_info).
RECORD_FIELDS = ("id", "status", "iterations", "objective", "rp", "rd", "gap", "seconds", "pivots_fixed",
                 "timeouts_recovered", "serial_launches", "setup_seconds", "solve_seconds", "teardown_seconds")
NF = len(RECORD_FIELDS)
_OPTIONAL = RECORD_FIELDS[9:]          # a custom solve_fn need not report these (0 then)
STATUS_INVALID_INPUT = -6.0          # IPM_ERR_INVALID_INPUT surfaced as a status
STATUS_ERROR = -99.0


def predicted_cost(m, n, nnz_col_sq=None, iters_est=40):
    """Predicted solve time (arbitrary units) for the LPT partition.  Measured on MI355X (profiles/r01_netlib_*.json):
    one iteration costs about 0.1 ms + 0.11 ms per 128-row block of the normal matrix from m = 27 to m = 16675 -- the
    blocked Cholesky is a latency chain, not a flop count -- so the model is linear in the block count.  (The cubic
    flop model this replaces put 11.2 s of a 12.5 s suite on one of two ranks.)  The iteration count is not
    predictable (14 ... the cap) and is taken as equal."""
    nblk = (int(m) + 127) // 128
    return iters_est * (0.1 + 0.11 * nblk)


def lpt_partition(costs, world):
    """Longest-processing-time greedy: sort by cost descending, give each item to the least
    loaded rank (ties -> lowest rank).  Deterministic, so every rank derives the same shards."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        shards[r].append(i)
        load[r] += float(costs[i])
    return shards


def _error_info(status=STATUS_ERROR):
    nan = float("nan")
    return dict(status=status, iterations=0, objective=nan, rp=nan, rd=nan, gap=nan, pivots_fixed=0)


def _row(i, info):
    """info dict of one solve -> its record row."""
    return [float(i), float(info["status"]), float(info["iterations"]), float(info["objective"]), float(info["rp"]),
            float(info["rd"]), float(info["gap"]), float(info["seconds"]), float(info["pivots_fixed"])] + \
           [float(info.get(k, 0.0)) for k in _OPTIONAL]


def _guarded(solve_fn, problem, **kw):
    """solve_fn(problem, **kw) -> info; an exception of any kind (library error, bad shapes, allocation failure) is
    turned into a STATUS_ERROR record so that the worker -- and with it the rank -- always reaches the collective."""
    try:
        return dict(solve_fn(problem, **kw))
    except Exception as e:
        import sys
        print("[batch] LP failed: %s: %s" % (type(e).__name__, e), file=sys.stderr, flush=True)
        return _error_info()


def solve_one(problem, device=0, tol=1e-8, max_iter=5000, y0=1.0, regularize=0.0, concurrent=False, start="reference",
              tol_gap=None):
    """Solve one LP (A, b, c) on `device` with the HIP path -> dict of statistics."""
    from . import _lib
    from .solver import solve_with_info
    A, b, c = problem
    t0 = time.perf_counter()
    try:
        _, _, _, info = solve_with_info(A, b, c, tol=tol, max_iter=max_iter, y0=y0, device=device,
                                        regularize=regularize, concurrent=concurrent, start=start, tol_gap=tol_gap)
        info = dict(info)
    except Exception as e:          # every failure becomes a record: the rank must still reach the all-gather
        import sys
        print("[batch] %d x %d LP failed: %s: %s" % (A.shape[0], A.shape[1], type(e).__name__, e), file=sys.stderr,
              flush=True)
        info = _error_info(STATUS_INVALID_INPUT if getattr(e, "code", None) == _lib.ERR_INVALID_INPUT else STATUS_ERROR)
    info["seconds"] = time.perf_counter() - t0
    return info


def _in_own_stream(solve_fn, problem, device, kw, stream=None):
    """Run solve_fn with a torch stream of its own as the thread's current stream (a handle binds to the current
    stream), so solves issued from different host threads overlap on the GPU.  (High-priority streams for the largest LPs
    were measured in round 4 and make the suite SLOWER: 14.5 LPs/s without, 13.0 / 13.3 / 11.2 with the LPs of >= 1000 /
    2000 / 3000 rows prioritised -- profiles/r04_netlib_stream_priority_rejected.txt.)"""
    try:
        import torch
        if torch.cuda.is_available():
            with torch.cuda.device(device), torch.cuda.stream(stream if stream is not None else torch.cuda.Stream(device=device)):
                return solve_fn(problem, device=device, **kw)
    except ImportError:
        pass
    return solve_fn(problem, device=device, **kw)


SMALL_ROWS = 1 << 30    # LPs up to this many rows share the GPU with each other (all of them; lower it to serialise the largest)


def solve_shard(problems, ids, device=0, solve_fn=solve_one, workers=1, **kw):
    """Solve problems[i] for i in ids on this rank's GPU -> (len(ids), NF) float64 records.

    workers > 1: the LPs are solved `workers` at a time from host threads, each handle on ONE stream of its own
    (IPM_FLAG_SINGLE_STREAM | IPM_FLAG_NO_DEVICE_POLLING): a mid-size LP is a latency chain that keeps a few CUs busy,
    so several of them overlap on one GPU -- provided their streams do not share hardware queues.  The HIP runtime has
    four; with the two-stream look-ahead of a lone solve two LPs already collide (2 x DEGEN3: 1.3x the time of one, 3 x:
    2.4x), with one stream per handle four overlap almost perfectly (4 x: 1.15x; tools/concurrency_probe.py) although
    each solve alone is 10-25 % slower without the look-ahead.  Results do not depend on the interleaving (every handle
    is independent and deterministic, and both schedules perform the same arithmetic)."""
    rec = np.zeros((len(ids), NF), dtype=np.float64)
    import queue
    free_streams = queue.Queue()        # the rank's worker streams, the same ones in every call (see _lockstep_streams)
    for st in _lockstep_streams(device, max(1, workers)):
        free_streams.put(st)

    def on_a_worker_stream(p, **k):
        st = free_streams.get()
        try:
            return _in_own_stream(solve_fn, p, device, k, stream=st)
        finally:
            free_streams.put(st)

    def one(row_i):
        row, i = row_i
        t0 = time.perf_counter()
        if workers > 1 and problems[i][0].shape[0] <= SMALL_ROWS:
            # handles that share the GPU must not poll on the device (IPM_FLAG_NO_DEVICE_POLLING, include/ipm_hip.h)
            info = _guarded(on_a_worker_stream, problems[i],
                            **(dict(kw, concurrent=True) if solve_fn is solve_one else kw))
        else:
            info = _guarded(solve_fn, problems[i], device=device, **kw)
        info.setdefault("seconds", time.perf_counter() - t0)
        rec[row] = _row(i, info)

    rows = list(enumerate(ids))
    if workers <= 1:
        for r in rows:
            one(r)
        return rec
    from concurrent.futures import ThreadPoolExecutor
    size = lambda r: problems[r[1]][0].shape[0]                     # noqa: E731
    for r in [r for r in rows if size(r) > SMALL_ROWS]:
        one(r)
    small = sorted([r for r in rows if size(r) <= SMALL_ROWS], key=lambda r: (-size(r), r[1]))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(one, small))
    return rec


# Tuning of solve_shard_lockstep (one MI355X, the 73 Netlib LPs; DESIGN.md 6 has the sweeps):
LOCKSTEP_MAX_ROWS = int(__import__("os").environ.get("IPM_LOCKSTEP_MAX_ROWS", 1 << 30))
# row limits of the size classes (one batch, one stream, one host thread per class; the last class is open ended).  Every step of a
# batch lasts as long as its slowest LP's kernel, so very different sizes in one batch inflate each other's chain
LOCKSTEP_CLASSES = [int(v) for v in __import__("os").environ.get("IPM_LOCKSTEP_CLASSES", "2200,3500").split(",") if v]
# host threads (= streams) for the LPs outside the batches (up to 128 rows, sparse factor), solved one after the other.  Batches +
# these should not be more than the four hardware queues HIP multiplexes its streams onto: a fifth stream costs 25 % of the suite
LOCKSTEP_CLASSIC_THREADS = int(__import__("os").environ.get("IPM_LOCKSTEP_CLASSIC_THREADS", 1))
# up to this many rows an LP takes the dense-tile factor (and joins a batch) even where a lone solve would take the sparse one
LOCKSTEP_DENSE_ROWS = int(__import__("os").environ.get("IPM_LOCKSTEP_DENSE_ROWS", 3500))


_LOCKSTEP_STREAMS = {}      # device -> the streams of solve_shard_lockstep, created ONCE back to back


def _lockstep_streams(device, n):
    """n torch streams for the batches and the one-at-a-time runners of a rank, the same ones in every call.  HIP maps a new
    stream onto the next of its four hardware queues in creation order: four streams created back to back sit on four
    different queues, whereas streams created per call (and per finished batch) drift onto occupied ones -- the second and third
    run of the suite in one process took 1.87 / 1.97 s instead of 1.41 (profiles/r04_netlib_repeated_runs_stream_reuse.txt)."""
    try:
        import torch
        if not torch.cuda.is_available():
            return [None] * n
    except ImportError:
        return [None] * n
    have = _LOCKSTEP_STREAMS.setdefault(int(device), [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=device))
    return have[:n]


def solve_shard_lockstep(problems, ids, device=0, workers=8, tol=1e-8, max_iter=5000, y0=1.0, regularize=0.0, tol_gap=None, **_):
    """Solve problems[i] for i in ids on this rank's GPU with the LOCKSTEP BATCH -> (len(ids), NF) records.

    `workers` host threads prepare the LPs (host analysis, handle, upload), largest first.  An LP of more than 128 rows on the
    dense-tile factor JOINS the running batch of its size class as soon as its handle is ready (ipm_batch_add between two steps):
    iteration k of all LPs of a class in the same launches (csrc/lockstep.h), so that the rank's chains of launches are those of
    the longest LP of each class instead of eight chains that share four hardware queues.  The LPs a batch cannot serve -- up to
    128 rows (fused single-workgroup kernel), the sparse multifrontal factor above LOCKSTEP_DENSE_ROWS rows -- are solved one after
    the other on LOCKSTEP_CLASSIC_THREADS streams, and by the class threads once their batch has finished.  Finished LPs are read
    back and destroyed by the pool while the batches run on.  Per-LP results are those of a one-at-a-time solve of the same
    handle, bit for bit (tests/test_gpu_lockstep.py)."""
    import queue
    from concurrent.futures import ThreadPoolExecutor
    from . import _lib
    from .solver import IpmSolver, LockstepBatch, _info, lockstep_eligible, prepare
    rec = np.zeros((len(ids), NF), dtype=np.float64)
    import threading
    classic_q = queue.Queue()

    def classic(problem, device=0, prepared=None):
        from .solver import solve_with_info
        A, b, c = problem
        _, _, _, info = solve_with_info(A, b, c, tol=tol, max_iter=max_iter, y0=y0, device=device, regularize=regularize,
                                        concurrent=True, tol_gap=tol_gap, prepared=prepared)
        return dict(info)

    def wants_lockstep(i):
        return 128 < problems[i][0].shape[0] <= LOCKSTEP_MAX_ROWS

    def one(row_i, ready):          # ready: queue of (row, lp id, solver, setup seconds) | None = a setup task ended without a handle
        row, i = row_i
        A, b, c = problems[i]
        t0 = time.perf_counter()
        sv, handed = None, False
        try:
            # host analysis ONCE: row order, factor path.  Up to LOCKSTEP_DENSE_ROWS rows the dense-tile factor even where a lone solve
            # would take the sparse one: inside a batch an LP whose program is shorter than the class leader's adds no launches
            prepared = prepare(A, b, c, factor=("dense" if wants_lockstep(i) and A.shape[0] <= LOCKSTEP_DENSE_ROWS else None))
            if wants_lockstep(i) and prepared.factor != "sparse":
                sv = IpmSolver(A, b, c, device=device, regularize=regularize, lockstep=True, concurrent=True, prepared=prepared)
                if lockstep_eligible(sv):
                    sv.init_state(y0)
                    ready.put((row, i, sv, time.perf_counter() - t0))
                    handed = True
                    return
                sv.close()
                sv = None
            classic_q.put((row, i, prepared, t0))         # solved by a classic runner thread (few streams: see LOCKSTEP_CLASSIC_THREADS)
            return
        except Exception as e:          # every failure becomes a record: the rank must still reach the all-gather
            import sys
            print("[batch] %d x %d LP failed: %s: %s" % (A.shape[0], A.shape[1], type(e).__name__, e), file=sys.stderr, flush=True)
            if sv is not None:
                sv.close()
            info = _error_info(STATUS_INVALID_INPUT if getattr(e, "code", None) == _lib.ERR_INVALID_INPUT else STATUS_ERROR)
        finally:
            if wants_lockstep(i) and not handed:
                ready.put(None)
        info.setdefault("seconds", time.perf_counter() - t0)
        rec[row] = _row(i, info)

    def classic_runner(own):            # own: ONE stream per runner for all its LPs (a new stream per LP walks through the hardware queues)
        while True:
            item = classic_q.get()
            if item is None:
                return
            row, i, prepared, t0 = item
            try:
                info = _in_own_stream(classic, problems[i], device, dict(prepared=prepared), stream=own)
            except Exception as e:
                import sys
                A = problems[i][0]
                print("[batch] %d x %d LP failed: %s: %s" % (A.shape[0], A.shape[1], type(e).__name__, e), file=sys.stderr, flush=True)
                info = _error_info(STATUS_INVALID_INPUT if getattr(e, "code", None) == _lib.ERR_INVALID_INPUT else STATUS_ERROR)
            info.setdefault("seconds", time.perf_counter() - t0)
            rec[row] = _row(i, info)

    def finish(item, t_join, t_done):
        row, i, sv, setup_s = item
        t3 = time.perf_counter()
        try:
            info = _info(sv)
            info["timeouts_recovered"] = sv.schedule()["timeouts_recovered"]
            info["serial_launches"] = 0
        except Exception:
            info = _error_info()
        sv.close()
        info["setup_seconds"], info["solve_seconds"], info["teardown_seconds"] = setup_s, t_done - t_join, time.perf_counter() - t3
        info["seconds"] = setup_s + (t_done - t_join)
        rec[row] = _row(i, info)

    # Size classes: ONE batch per class, each on a stream of its own, run from a host thread of its own.  In a batch every global step
    # lasts as long as its slowest LP's kernel, so LPs of very different size in one batch inflate each other's chain (measured: all 53
    # LPs in one batch run 5.4 ms per iteration where the largest alone needs 2.5); a class of similar LPs keeps the chain at its leader's.
    def cls_of(i):
        m = problems[i][0].shape[0]
        for k, lim in enumerate(LOCKSTEP_CLASSES):
            if m <= lim:
                return k
        return len(LOCKSTEP_CLASSES)
    ncls = len(LOCKSTEP_CLASSES) + 1
    ready = [queue.Queue() for _ in range(ncls)]

    def one_cls(row_i):
        k = cls_of(row_i[1])
        one(row_i, ready[k])

    # the lockstep candidates first (largest first: a batch should start with its longest chains), then the rest
    rows = sorted(enumerate(ids), key=lambda r: (not wants_lockstep(r[1]), -problems[r[1]][0].shape[0], r[1]))
    expected = [0] * ncls
    for r in rows:
        if wants_lockstep(r[1]):
            expected[cls_of(r[1])] += 1
    pool = ThreadPoolExecutor(max_workers=max(1, workers))
    futs = [pool.submit(one_cls, r) for r in rows]
    tails, tails_lock = [], threading.Lock()

    def run_class(k):
        arrived, joined = 0, {}
        if expected[k] == 0:
            return
        try:
            with LockstepBatch(device=device, tol=tol, max_iter=max_iter, tol_gap=tol_gap, stream=streams[k]) as lb:
                while arrived < expected[k] or lb.active > 0:
                    # every handle that is ready joins now; with nothing running, wait for the next one
                    while arrived < expected[k]:
                        try:
                            item = ready[k].get(block=(lb.active == 0))
                        except queue.Empty:
                            break
                        arrived += 1
                        if item is not None:
                            joined[id(item[2])] = (item, time.perf_counter())
                            lb.add(item[2])
                    if lb.active == 0:
                        continue
                    done = lb.step()
                    t_done = time.perf_counter()
                    for sv in done:
                        item, t_join = joined.pop(id(sv))
                        with tails_lock:
                            tails.append(pool.submit(finish, item, t_join, t_done))
        except Exception as e:          # the handles still in the batch become error records: the rank must still reach the all-gather
            import sys
            print("[batch] lockstep batch failed: %s: %s" % (type(e).__name__, e), file=sys.stderr, flush=True)
            for item, _t in joined.values():
                rec[item[0]] = _row(item[1], dict(_error_info(), seconds=0.0))
                try:
                    item[2].close()
                except Exception:
                    pass

    def run_class_then_classic(k):
        run_class(k)
        if expected[k] and __import__("os").environ.get("IPM_LOCKSTEP_STEAL", "1") != "0":
            classic_runner(streams[k])  # its batch is finished, its stream idle: help with the LPs outside the batches

    nclassic = max(1, LOCKSTEP_CLASSIC_THREADS)
    streams = _lockstep_streams(device, ncls + nclassic)
    runners = [threading.Thread(target=run_class_then_classic, args=(k,)) for k in range(ncls)]
    crunners = [threading.Thread(target=classic_runner, args=(streams[ncls + j],)) for j in range(nclassic)]
    for t in runners + crunners:
        t.start()
    for f in futs:
        f.result()                       # every LP is set up: the classic queue is complete
    for _ in crunners + runners:
        classic_q.put(None)
    for t in runners + crunners:
        t.join()
    for f in tails:
        f.result()
    pool.shutdown()
    return rec


def gather_records(local, shard_sizes, dist=None, device=None):
    """All ranks obtain all records, ordered by LP id.  One all_gather of (max_shard, NF) float64
    tensors (padded with id = -1); 8 NF = 112 B per LP, latency-bound, xGMI bandwidth irrelevant."""
    if dist is None or not dist.is_initialized():
        out = local
    else:
        import torch
        world = dist.get_world_size()
        cap = max(max(shard_sizes), 1)
        pad = np.full((cap, NF), -1.0)
        pad[:local.shape[0]] = local
        t = torch.from_numpy(pad)
        if device is not None:
            t = t.to(device)
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        out = np.concatenate([b.cpu().numpy()[:shard_sizes[r]] for r, b in enumerate(bufs)], axis=0)
    return out[np.argsort(out[:, 0], kind="stable")]


_CALLS = 0          # run_batch() calls so far: names the shared work counter of a call (same sequence on every rank)


def make_store(rank, world, host=None, port=None, timeout_s=300):
    """A TCP key-value store for the self-scheduling counter (public torch.distributed.TCPStore API; rank 0 hosts it).
    Control plane only: one small round trip per LP.  The caller may hand the same store to init_process_group."""
    import datetime
    import os
    import torch.distributed as tdist
    host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port if port is not None else int(os.environ.get("MASTER_PORT", "29500")) + 1)
    return tdist.TCPStore(host, port, world, is_master=(rank == 0), timeout=datetime.timedelta(seconds=timeout_s),
                          wait_for_workers=True)


def _solve_dynamic(problems, order, store, key, device, solve_fn, workers, **kw):
    """Self-scheduling: every worker thread of every rank takes the next LP of `order` (most expensive first) from
    an atomic counter on the rendezvous store.  Iteration counts (14 ... the cap) make a static partition lose up to
    40 % to imbalance; the counter costs one small TCP round trip per LP.  Returns an (n, NF) table whose rows not
    solved on this rank keep id = -1."""
    import threading
    n = len(problems)
    rec = np.full((n, NF), -1.0)
    lock = threading.Lock()

    def take():
        with lock:
            return int(store.add(key, 1)) - 1

    streams = _lockstep_streams(device, max(1, workers))      # one fixed stream per worker loop, the same ones in every call

    def loop(w):
        while True:
            j = take()
            if j >= n:
                return
            i = order[j]
            t0 = time.perf_counter()
            if workers > 1:
                info = _guarded(lambda p, **k: _in_own_stream(solve_fn, p, device, k, stream=streams[w]), problems[i],
                                **(dict(kw, concurrent=True) if solve_fn is solve_one else kw))
            else:
                info = _guarded(solve_fn, problems[i], device=device, **kw)
            info.setdefault("seconds", time.perf_counter() - t0)
            rec[i] = _row(i, info)

    if workers <= 1:
        loop(0)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(loop, range(workers)))
    return rec


def _gather_sparse(rec, dist, device=None):
    """One all_gather of the (n, NF) tables; row i is taken from the rank that solved LP i."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(rec))
    if device is not None:
        t = t.to(device)
    bufs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, t)
    out = np.full(rec.shape, -1.0)
    for b in bufs:
        a = b.cpu().numpy()
        own = a[:, 0] >= 0
        out[own] = a[own]
    return out


# "auto": the lockstep batches pay off when a GPU holds many more LPs of the dense-tile size range than the four hardware queues can
# overlap as separate chains (73-LP suite, 61 such LPs on one GPU: 23.5 LPs/s against 14.7 one-at-a-time); with few of them the
# chains of a batch only wait for each other (26-LP parity set, 17 such LPs: 86-123 LPs/s against 116-140) -- sweeps in
# profiles/r04_netlib_lockstep_vs_classic_sweep.txt.  The rule counts LPs of more than 128 rows per rank.
LOCKSTEP_MIN_LPS = int(__import__("os").environ.get("IPM_LOCKSTEP_MIN_LPS", 24))


def lockstep_wanted(problems, world=1, workers=8, mode="auto"):
    """Whether run_batch(lockstep=mode) uses the lockstep batches: the same answer on every rank."""
    if mode in (False, None, 0, "0"):
        return False
    if workers <= 1:
        return False
    if mode == "auto":
        return sum(1 for p in problems if p[0].shape[0] > 128) >= LOCKSTEP_MIN_LPS * max(1, world)
    return True


def run_batch(problems, costs=None, device=0, dist=None, gather_device=None, solve_fn=solve_one, workers=1,
              schedule="static", store=None, collective_at_world_one=False, lockstep=False, **kw):
    """Shard `problems` (list of (A, b, c)) over the ranks of `dist`, solve, gather statistics.

    schedule="static": deterministic LPT partition on the predicted cost, no scheduling traffic at all.
    schedule="dynamic" (N > 1): the ranks pull LPs, most expensive first, from a counter on `store` (a
    torch.distributed store shared by all ranks, e.g. make_store(); without one the schedule is "static").  Either way the only collective is
    ONE all-gather of the statistics records.  Returns (records sorted by id, this rank's wall seconds).  Without an
    initialised process group this is the single-GPU loop; so it is with a group of ONE rank, unless
    collective_at_world_one asks for the distributed code path anyway (store counter, device tensors through
    dist.all_gather): that is how the RCCL branch is exercised on a one-GPU box (tests/test_gpu_parity.py)."""
    global _CALLS
    _CALLS += 1
    live = dist is not None and dist.is_initialized()
    world = dist.get_world_size() if live else 1
    multi = world > 1 or (live and collective_at_world_one)
    rank = dist.get_rank() if live else 0
    if costs is None:
        costs = [predicted_cost(p[0].shape[0], p[0].shape[1]) for p in problems]
    # lockstep=True: each rank solves its shard of the static partition with the lockstep batch (solve_shard_lockstep): the ranks'
    # wall times are then set by their longest LP, which a pull-based schedule cannot improve
    # (lockstep="auto": only where a rank holds enough LPs for it, see LOCKSTEP_MIN_LPS)
    lockstep = solve_fn is solve_one and lockstep_wanted(problems, world, workers, lockstep)
    store = store if (multi and schedule == "dynamic" and not lockstep) else None
    t0 = time.perf_counter()
    if store is not None:
        order = sorted(range(len(problems)), key=lambda i: (-float(costs[i]), i))
        local = _solve_dynamic(problems, order, store, "ipm_batch_next_%d" % _CALLS, device, solve_fn, workers, **kw)
        seconds = time.perf_counter() - t0
        return _gather_sparse(local, dist, device=gather_device), seconds
    shards = lpt_partition(costs, world)
    if lockstep:
        local = solve_shard_lockstep(problems, shards[rank], device=device, workers=workers, **{k: v for k, v in kw.items() if k != "start"})
    else:
        local = solve_shard(problems, shards[rank], device=device, solve_fn=solve_fn, workers=workers, **kw)
    seconds = time.perf_counter() - t0
    records = gather_records(local, [len(s) for s in shards], dist=dist if multi else None,
                             device=gather_device)
    return records, seconds


def summarize(records):
    """Convergence statistics of a gathered batch (the reduction the north star asks for)."""
    status = records[:, 1]
    conv = status == 1.0
    return dict(n=int(records.shape[0]), converged=int(conv.sum()), max_iter=int((status == 2.0).sum()),
                nan=int((status == 3.0).sum()), invalid=int((status == STATUS_INVALID_INPUT).sum()),
                errors=int((status == STATUS_ERROR).sum()),
                total_iterations=int(records[:, 2].sum()), solve_seconds_sum=float(records[:, 7].sum()),
                pivots_fixed=int(records[:, 8].sum()),
                timeouts_recovered=int(records[:, 9].sum()), serial_launches=int(records[:, 10].sum()),
                setup_seconds_sum=float(records[:, 11].sum()), device_solve_seconds_sum=float(records[:, 12].sum()),
                teardown_seconds_sum=float(records[:, 13].sum()))
