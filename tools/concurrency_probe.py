#!/usr/bin/env python3
"""How well do N independent solves of the same mid-size LP overlap on one GPU when driven from N host threads (each on its
own stream)?  Prints the wall time of N concurrent solves relative to one.   python tools/concurrency_probe.py [NAME]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import interiorpointmethod_amd as ipm
from interiorpointmethod_amd.matio import load_npz_problem
name = sys.argv[1] if len(sys.argv) > 1 else "DEGEN3"
A, b, c, cTlb, valid = load_npz_problem(os.path.join(ROOT, "tests", "golden", "netlib", name + ".npz"))
ipm.solve_with_info(A, b, c, tol=1e-8, max_iter=20)
def one(out, i, barrier, concurrent):
    with torch.cuda.stream(torch.cuda.Stream()):
        with ipm.IpmSolver(A, b, c, concurrent=concurrent) as sv:
            sv.init_state(1.0)
            barrier.wait()
            t0 = time.perf_counter()
            st = sv.solve(tol=1e-8, max_iter=300)
            out[i] = (time.perf_counter() - t0, st["iterations"])
            barrier.wait()
base = None
for n in (1, 2, 3, 4, 6):
    out = [None] * n
    bar = threading.Barrier(n)
    ts = [threading.Thread(target=one, args=(out, i, bar, n > 1)) for i in range(n)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    mx = max(o[0] for o in out)
    if base is None: base = mx
    print("%s x %d concurrent: slowest solve %.3f s (%.2fx of one), aggregate %.1f iterations/ms" % (name, n, mx, mx / base, sum(o[1] for o in out) / mx / 1e3))
