"""Batched-LP mode: independent LP instances sharded across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for
tests).  An LP is solved entirely on one GPU -- there is no exchange inside a solve -- so the
only collective is ONE all-gather of a fixed-size statistics record per LP at the end
(SURVEY.md 8e).  The partition is a deterministic LPT-greedy assignment on a predicted cost,
computed identically on every rank, so no scheduling traffic is needed either.

The reference has no counterpart (it is single process); the loop being distributed is the
driver loop of script.py:147-173 over the Netlib files.
"""
from __future__ import annotations

import time

import numpy as np

RECORD_FIELDS = ("id", "status", "iterations", "objective", "rp", "rd", "gap", "seconds", "pivots_fixed")
NF = len(RECORD_FIELDS)
STATUS_INVALID_INPUT = -6.0          # IPM_ERR_INVALID_INPUT surfaced as a status
STATUS_ERROR = -99.0


def predicted_cost(m, n, nnz_col_sq=None, iters_est=40):
    """iters * (contraction flops + m^3/3): dense-B factorization dominates."""
    form = float(m) * m * n if nnz_col_sq is None else float(nnz_col_sq)
    return iters_est * (form + float(m) ** 3 / 3.0)


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


def solve_one(problem, device=0, tol=1e-8, max_iter=5000, y0=1.0, regularize=0.0):
    """Solve one LP (A, b, c) on `device` with the HIP path -> dict of statistics."""
    from . import _lib
    from .solver import solve_with_info
    A, b, c = problem
    t0 = time.perf_counter()
    try:
        _, _, _, info = solve_with_info(A, b, c, tol=tol, max_iter=max_iter, y0=y0, device=device,
                                        regularize=regularize)
        info = dict(info)
    except _lib.IpmError as e:
        nan = float("nan")
        info = dict(status=STATUS_INVALID_INPUT if e.code == _lib.ERR_INVALID_INPUT else STATUS_ERROR,
                    iterations=0, objective=nan, rp=nan, rd=nan, gap=nan, pivots_fixed=0)
    info["seconds"] = time.perf_counter() - t0
    return info


def solve_shard(problems, ids, device=0, solve_fn=solve_one, **kw):
    """Solve problems[i] for i in ids on this rank's GPU -> (len(ids), NF) float64 records."""
    rec = np.zeros((len(ids), NF), dtype=np.float64)
    for row, i in enumerate(ids):
        t0 = time.perf_counter()
        info = dict(solve_fn(problems[i], device=device, **kw))
        info.setdefault("seconds", time.perf_counter() - t0)
        rec[row] = [float(i), float(info["status"]), float(info["iterations"]), float(info["objective"]),
                    float(info["rp"]), float(info["rd"]), float(info["gap"]), float(info["seconds"]),
                    float(info["pivots_fixed"])]
    return rec


def gather_records(local, shard_sizes, dist=None, device=None):
    """All ranks obtain all records, ordered by LP id.  One all_gather of (max_shard, NF) float64
    tensors (padded with id = -1); ~72 B per LP, latency-bound, xGMI bandwidth irrelevant."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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


def run_batch(problems, costs=None, device=0, dist=None, gather_device=None, solve_fn=solve_one, **kw):
    """Shard `problems` (list of (A, b, c)) over the ranks of `dist`, solve, gather statistics.

    Returns (records sorted by id, this rank's wall seconds).  Without an initialised process
    group this is the single-GPU loop."""
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    if costs is None:
        costs = [predicted_cost(p[0].shape[0], p[0].shape[1]) for p in problems]
    shards = lpt_partition(costs, world)
    t0 = time.perf_counter()
    local = solve_shard(problems, shards[rank], device=device, solve_fn=solve_fn, **kw)
    seconds = time.perf_counter() - t0
    records = gather_records(local, [len(s) for s in shards], dist=dist if world > 1 else None,
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
                pivots_fixed=int(records[:, 8].sum()))
