#!/usr/bin/env python3
"""bench.py -- IPM iterations/sec of the HIP Newton/KKT path on the dense synthetic LP.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full Mehrotra predictor-corrector iteration (stop
test, form A D^2 A^T, Cholesky, two solves, two ratio tests, update) on the BASELINE.json
config "dense synthetic LP m=4096 n=8192 fp64" (generator of SURVEY.md 8d, seed 0, start
x=s=1, y=0).  All inputs are resident in HBM before the timed region.  The iterate is reset
to the start point every 20 steps (3 tiny fill launches, inside the timed region) so that
any K stays on the pre-convergence trajectory (the LP converges in 24 iterations).

N > 1 (launched by torch.distributed.run, one rank per GPU): the single-LP dense path does
not shard ("replicas only"): every rank solves its own replica, no data-path
collective; value = N*K / max-over-ranks time  (weak scaling).

Prints ONE JSON line (rank 0) with the contract keys plus `roofline` (the dominant kernel,
the fp64-MFMA A D^2 A^T contraction, timed with HIP events on the solver's stream inside the
timed region) and `cpu_baseline` (the NumPy normal-equations oracle on the host cores,
bounded sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_DEFAULT, N_DEFAULT = 4096, 8192
RESET_EVERY = 20
PEAK_FP64_MFMA_TFLOPS = 78.6      # MI355X dense fp64 matrix peak


def cpu_baseline(A, b, c, budget_s=20.0, max_its=6):
    """Oracle (NumPy normal equations + LAPACK Cholesky) timed on the host: bounded sample."""
    import numpy as np
    from oracle import ipm_oracle as O
    m, n = A.shape
    x, y, s = O.initial_point(m, n, 0.0)
    t0 = time.perf_counter()
    its = 0
    while its < max_its and (time.perf_counter() - t0) < budget_s:
        x, y, s, _ = O.iterate(A, b, c, x, y, s, method="normal")
        its += 1
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": its / dt, "unit": "iterations/s", "cores": int(threads), "kind": "port",
            "sample": "%d iterations of oracle.iterate(method='normal') (NumPy (A*d)@A.T + LAPACK potrf) "
                      "on the same %dx%d LP from the start point, %.1f s, host has %d logical CPUs" % (
                          its, m, n, dt, os.cpu_count() or 0)}


def _dist_setup(local_rank, world):
    """(dist module or None, device index, device for the scalar reductions).  One rank per GPU over RCCL; the
    environment IPM_BENCH_BACKEND=gloo + IPM_BENCH_ONE_DEVICE=1 rehearses the multi-rank code path on a one-GPU box
    (every rank on device 0, CPU collectives)."""
    import torch
    if world <= 1:
        return None, local_rank, "cuda"
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("IPM_BENCH_BACKEND", "nccl")
    dev = 0 if os.environ.get("IPM_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        return dist, dev, "cuda"
    dist.init_process_group(backend=backend)
    return dist, dev, "cpu"


PARITY_SET = ["AFIRO", "BANDM", "DEGEN2", "E226", "FIT1P", "GROW15", "GROW22", "GROW7", "KB2", "MAROS-R7", "SC105",
              "SC205", "SC50A", "SC50B", "SCSD1", "SCSD6", "SCSD8", "SCTAP1", "SCTAP2", "SCTAP3", "SHARE2B",
              "STOCFOR1", "STOCFOR2", "STOCFOR3", "TRUSS", "WOODW"]


def netlib_main(args):
    """Batched-LP mode (BASELINE.json configs[3]): the Netlib fixtures sharded over the ranks, one LP per
    GPU at a time, a single RCCL all-gather of the statistics records at the end."""
    import glob
    import numpy as np
    import torch
    from interiorpointmethod_amd import batch
    from interiorpointmethod_amd.matio import load_npz_problem

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist, dev, red_dev = _dist_setup(local_rank, world)
    names, probs, costs = [], [], []
    if args.netlib_set == "general":
        # the reference's benchmarks_full files (general form) through the front end: conversion on the host, untimed
        from scipy import sparse
        from interiorpointmethod_amd import general_form as G
        for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "general", "*.npz"))):
            z = np.load(f)

            def mat(prefix):
                if prefix + "_none" in z.files:
                    return None
                return sparse.csc_matrix((z[prefix + "_data"], z[prefix + "_indices"], z[prefix + "_indptr"]),
                                         shape=tuple(int(v) for v in z[prefix + "_shape"]))
            A, b, c, _ = G.standard_form(z["c"], Aeq=mat("Aeq"), beq=z["beq"] if "beq" in z.files else None,
                                         Aineq=mat("Aineq"), bineq=z["bineq"] if "bineq" in z.files else None,
                                         lb=z["lb"], ub=z["ub"])
            if A.shape[0] > args.max_m:
                continue
            names.append(os.path.basename(f)[:-4])
            probs.append((sparse.csc_matrix(A), b, c))
            costs.append(batch.predicted_cost(A.shape[0], A.shape[1]))
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "netlib", "*.npz")) if args.netlib_set != "general" else []):
        nm = os.path.basename(f)[:-4]
        if args.netlib_set == "parity" and nm not in PARITY_SET:
            continue
        A, b, c, cTlb, valid = load_npz_problem(f)
        if not valid or A.shape[0] > args.max_m:
            continue
        names.append(nm)
        probs.append((A, b, c))
        colnnz = np.diff(A.indptr).astype(np.float64)
        costs.append(batch.predicted_cost(A.shape[0], A.shape[1]))     # dense-A contraction today
    # warm-up: one small solve per rank (library load, first-launch costs) outside the timed region
    batch.solve_one(probs[names.index("AFIRO")] if "AFIRO" in names else probs[0], device=dev)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    rec, _ = batch.run_batch(probs, costs=costs, device=dev, dist=dist,
                             gather_device=torch.device("cuda", dev) if (dist is not None and red_dev == "cuda") else None,
                             tol=1e-8, regularize=args.regularize, workers=args.workers,
                             schedule=args.schedule, start=args.start,
                             # the general-form driver's own settings: e3 = 1e-6, at most 999 iterations (main.py:1088-1127)
                             **(dict(max_iter=999, tol_gap=1e-6) if args.netlib_set == "general" else dict(max_iter=300)))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        summ = batch.summarize(rec)
        out = {"metric": "Netlib LPs/sec (benchmarks_full/ general-form suite, batched, tol=1e-8, e3=1e-6, cap 999)"
                         if args.netlib_set == "general" else "Netlib LPs/sec (benchmarks/ suite, batched, tol=1e-8, cap 300)",
               "value": summ["converged"] / elapsed, "unit": "LPs/s", "n_gpus": max(world, 1), "steps": len(names),
               "warmup": 1, "ms_per_step": 1e3 * elapsed / max(len(names), 1), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "netlib fixtures (tests/golden/netlib)",
               "config": {"workload": "Netlib %s set, %d LPs over %d GPU(s), %s; per GPU %d LP(s) in flight on separate streams" % (
                   args.netlib_set, len(names), max(world, 1),
                   "one rank" if world <= 1 else ("self-scheduled from the rendezvous store" if args.schedule == "dynamic"
                                                  else "static LPT partition"), max(1, args.workers))},
               "summary": summ, "wall_seconds": elapsed, "regularize": args.regularize, "start_point": args.start,
               "per_lp": {names[int(r[0])]: {"status": int(r[1]), "it": int(r[2]), "obj": r[3], "s": round(r[7], 3)}
                          for r in rec}}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--m", type=int, default=M_DEFAULT)
    ap.add_argument("--n", type=int, default=N_DEFAULT)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="dense", choices=["dense", "netlib"],
                    help="dense: IPM iterations/s on the synthetic LP (default, the headline metric); "
                         "netlib: LPs/s over the committed Netlib fixtures, sharded over the ranks")
    ap.add_argument("--netlib-set", default="all", choices=["all", "parity", "general"],
                    help="all 73 valid standard-form files, the 26 on which the reference converges, or the 72 general-form "
                         "files (benchmarks_full) through the general-form front end")
    ap.add_argument("--max-m", type=int, default=1 << 30, help="netlib: skip LPs with more rows")
    ap.add_argument("--workers", type=int, default=2, help="netlib: small LPs solved concurrently per GPU (1 = strictly one at a time)")
    ap.add_argument("--schedule", default="dynamic", choices=["dynamic", "static"],
                    help="netlib, N > 1: pull LPs from a shared counter (rendezvous store) or static LPT partition")
    ap.add_argument("--start-point", dest="start", default="reference", choices=["reference", "mehrotra"],
                    help="netlib: start point; reference = x=s=y=1 (sparse_interior.py:193-200, parity mode), mehrotra = "
                         "Mehrotra's least-squares start (optional mode, not the reference's algorithm)")
    ap.add_argument("--regularize", type=float, default=0.0, help="netlib: Tikhonov shift (0 = reference-faithful)")
    args = ap.parse_args()
    if args.workload == "netlib":
        return netlib_main(args)

    import numpy as np
    import torch
    import interiorpointmethod_amd as ipm
    from interiorpointmethod_amd.workloads import synthetic_lp, flops_per_iteration

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist, dev, red_dev = _dist_setup(local_rank, world)
    ngpu = max(world, 1)
    if args.gpus != ngpu and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    m, n = args.m, args.n
    A, b, c = synthetic_lp(m, n, seed=0)              # same LP on every rank (replicas)
    sv = ipm.IpmSolver(A, b, c, device=dev)

    def run(steps, profile=0):
        sv.set_profiling(profile)
        done, total_ms = 0, 0.0
        form_ms = factor_ms = tri_ms = other_ms = 0.0
        while done < steps:
            k = min(RESET_EVERY, steps - done)
            sv.init_state(0.0)
            st = sv.iterate(k)
            total_ms += st["solve_ms"]
            if profile:
                ph = sv.phase_ms()
                form_ms += ph["form"] * k; factor_ms += ph["factor"] * k
                tri_ms += ph["trisolve"] * k; other_ms += ph["other"] * k
            done += k
        return total_ms, (form_ms, factor_ms, tri_ms, other_ms), st

    run(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    dev_ms, phases, st = run(args.steps, profile=1)      # two event records per step around the dominant kernel
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # phase breakdown from a separate, untimed pass (nine event records per step cost ~1 %)
    _, phases_all, _ = run(min(args.steps, RESET_EVERY), profile=2)
    torch.cuda.synchronize()
    KB = min(args.steps, RESET_EVERY)

    if rank == 0:
        K = args.steps
        its_per_s = ngpu * K / elapsed
        form_ms = phases[0] / K
        flops_form = float(m) * m * n                       # lower-triangle SYRK, SURVEY 8(d)
        achieved = flops_form / (form_ms * 1e-3) / 1e12 if form_ms > 0 else 0.0
        traffic = None          # HBM bytes per launch of the dominant kernel: PMC pass collected separately
        try:                    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/r01_pmc_form_kernel.json)
            if (m, n) == (M_DEFAULT, N_DEFAULT):
                with open(os.path.join(ROOT, "profiles", "r01_pmc_form_kernel.json")) as fh:
                    traffic = json.load(fh)["derived"]["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "IPM iterations/sec (m=%d,n=%d dense LP)" % (m, n),
            "value": its_per_s, "unit": "iterations/s", "n_gpus": ngpu, "steps": K, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "dense synthetic LP m=%d n=%d fp64, seed 0, start x=s=1 y=0 "
                                   "(BASELINE.json configs[1]); replicas per GPU" % (m, n),
                       "reset_every": RESET_EVERY},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "traffic_note": "L2<->fabric bytes per launch from the committed PMC pass (2*FETCH_SIZE + WRITE_SIZE, "
                                         "gfx950 correction); 0.34 GB algorithmic; with eight private L2s the floor for this "
                                         "tiling is ~1.1 GB (each XCD reads half the row panels of A); part of it is served by "
                                         "the 256 MB Infinity Cache, so HBM bytes are lower; <1 TB/s either way: MFMA-bound",
                         "kernel": "gemm_nt_f64_kernel<128,128,16,2,2> (B = A diag(d) A^T)",
                         "flops_per_launch": flops_form, "avg_launch_ms": form_ms},
            "phases_ms_per_step": {"form": form_ms, "factor": phases_all[1] / KB, "trisolve": phases_all[2] / KB,
                                   "other": phases_all[3] / KB, "device_total": dev_ms / K,
                                   "note": "form and device_total from the timed region; the rest from an untimed pass"},
            "whole_iteration": {"flops_per_iteration": flops_per_iteration(m, n),
                                "tflops": flops_per_iteration(m, n) * its_per_s / ngpu / 1e12,
                                "frac_of_fp64_mfma_peak": flops_per_iteration(m, n) * its_per_s / ngpu / 1e12 / PEAK_FP64_MFMA_TFLOPS},
            "objective_after_last_block": st["objective"],
        }
        # guard against a fast-but-wrong kernel variant: 20 iterations from the start point of the default LP give
        # c^T x = -376.12529405 (the LP converges to -376.1254474176 after 24); checked when the last block is a full one
        if (m, n) == (M_DEFAULT, N_DEFAULT) and args.steps % RESET_EVERY == 0:
            ok = abs(st["objective"] - (-376.12529405)) <= 1e-6 * 376.0
            out["objective_check"] = "ok" if ok else "MISMATCH"
            if not ok:
                print("bench: objective after %d iterations is %.12e, expected -3.7612529405e+02" % (RESET_EVERY, st["objective"]),
                      file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, b, c)
        print(json.dumps(out))
    sv.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
