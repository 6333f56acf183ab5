#!/usr/bin/env python3
"""Stage-by-stage GPU diagnostic of the HIP path against NumPy / the oracle.

Not a test (tests/ holds those): a verbose bring-up tool that keeps going after a failing
stage so one GPU call reports on every kernel.  Usage: python tools/gpu_diag.py [stage ...]
"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interiorpointmethod_amd as ipm  # noqa: E402
from interiorpointmethod_amd.matio import load_npz_problem  # noqa: E402
from oracle import ipm_oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def stage_form():
    for (m, n) in ((16, 32), (200, 300), (130, 1000), (1000, 2000)):
        rng = np.random.default_rng(m)
        A = rng.standard_normal((m, n))
        d = rng.uniform(0.1, 10.0, n)
        with ipm.IpmSolver(A, np.zeros(m), np.zeros(n)) as sv:
            B = sv.form_normal_matrix(d)
        ref = (A * d) @ A.T
        print("form %4dx%-5d relerr=%.3e  sym=%.1e" % (m, n, relerr(B, ref), float(np.max(np.abs(B - B.T)))))
    # asymmetric integer data: exact check of the MFMA lane maps
    m, n = 40, 24
    A = np.arange(m * n, dtype=np.float64).reshape(m, n) % 7 - 3
    A[3, 5] = 11
    d = np.arange(1, n + 1, dtype=np.float64)
    with ipm.IpmSolver(A, np.zeros(m), np.zeros(n)) as sv:
        B = sv.form_normal_matrix(d)
    print("form exact-int max|diff| =", float(np.max(np.abs(B - (A * d) @ A.T))))


def stage_chol():
    for m in (16, 100, 128, 129, 300, 700, 1500):
        rng = np.random.default_rng(m)
        M = rng.standard_normal((m, m + 10))
        B = M @ M.T + 0.1 * np.eye(m)
        rhs = rng.standard_normal(m)
        with ipm.IpmSolver(np.eye(m, 1), np.zeros(m), np.zeros(1)) as sv:
            z, nfix = sv.solve_linear(B, rhs)
            L = sv.get_factor()
        Lref = np.linalg.cholesky(B)
        zref = np.linalg.solve(B, rhs)
        print("chol m=%4d  L relerr=%.3e  solve relerr=%.3e  resid=%.3e fixed=%d" % (
            m, relerr(L, Lref), relerr(z.ravel(), zref),
            float(np.linalg.norm(B @ z.ravel() - rhs) / np.linalg.norm(rhs)), nfix))
    # guard: rank-deficient matrix
    m = 200
    rng = np.random.default_rng(7)
    M = rng.standard_normal((m, 150))
    B = M @ M.T
    rhs = B @ rng.standard_normal(m)
    with ipm.IpmSolver(np.eye(m, 1), np.zeros(m), np.zeros(1)) as sv:
        z, nfix = sv.solve_linear(B, rhs)
    print("guard rank-def m=200 rank=150: fixed=%d finite=%s resid=%.3e" % (
        nfix, bool(np.all(np.isfinite(z))), float(np.linalg.norm(B @ z.ravel() - rhs) / np.linalg.norm(rhs))))


def stage_direction():
    for name in ("AFIRO", "SC50A", "BANDM"):
        z = np.load(os.path.join(G, "kat_%s.npz" % name))
        m, n = (int(v) for v in z["shape"])
        from scipy import sparse
        A = sparse.csc_matrix((z["A_data"], z["A_indices"], z["A_indptr"]), shape=(m, n))
        with ipm.IpmSolver(A, z["b"], z["c"]) as sv:
            for k in z["iters"]:
                pre = "k%d_" % int(k)
                sv.set_state(z[pre + "x"], z[pre + "y"], z[pre + "s"])
                dxa, dya, dsa = sv.newton_direction(False)
                st = dict(sv.stats)
                dx, dy, ds = sv.newton_direction(True)
                st2 = dict(sv.stats)
                print("dir %-6s k=%-3d pred: dx %.2e dy %.2e ds %.2e | a_aff %.3e/%.3e (ref %.3e/%.3e) | "
                      "sigma %.6e (ref %.6e) | corr: dx %.2e dy %.2e ds %.2e | fixed %d" % (
                          name, int(k), relerr(dxa, z[pre + "dxa"]), relerr(dya, z[pre + "dya"]),
                          relerr(dsa, z[pre + "dsa"]), st["alpha_aff_p"], st["alpha_aff_d"],
                          float(z[pre + "alpha_aff_p"]), float(z[pre + "alpha_aff_d"]), st2["sigma"],
                          float(z[pre + "sigma"]), relerr(dx, z[pre + "dx"]), relerr(dy, z[pre + "dy"]),
                          relerr(ds, z[pre + "ds"]), st2["pivots_fixed"]))


def stage_solve():
    for nm in ("ex1", "ex2", "ex3", "syn_64x128", "syn_256x512", "syn_512x1024"):
        z = np.load(os.path.join(G, "dense_%s.npz" % nm))
        if nm.startswith("syn"):
            m, n = (int(v) for v in z["shape"])
            A, b, c = O.synthetic_lp(m, n)
        else:
            A, b, c = z["A"], z["b"], z["c"]
        t0 = time.time()
        x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, y0=0.0, max_iter=500)
        print("solve %-13s it=%3d (ref %3d) obj=%.12e (ref %.12e) rel=%.2e rp=%.1e rd=%.1e gap=%.1e st=%s fixed=%d %.2fs" % (
            nm, info["iterations"], int(z["iterations"]), info["objective"], float(z["objective"]),
            abs(info["objective"] - float(z["objective"])) / max(1, abs(float(z["objective"]))),
            info["rp"], info["rd"], info["gap"], info["status_name"], info["pivots_fixed"], time.time() - t0))
    for name in ("AFIRO", "SC50A", "SC50B", "BANDM", "SC105", "SC205", "E226", "KB2", "SHARE2B", "STOCFOR1",
                 "SCSD1", "SCTAP1", "DEGEN2", "GROW7", "SCSD6"):
        e = np.load(os.path.join(G, "e2e_%s.npz" % name))
        A, b, c, cTlb, valid = load_npz_problem(os.path.join(G, "netlib", name + ".npz"))
        t0 = time.time()
        x, y, s, info = ipm.solve_with_info(A, b, c, tol=1e-8, y0=1.0, max_iter=500)
        print("solve %-9s it=%3d (ref %3d) obj=%.12e (ref %.12e) rel=%.2e rp=%.1e rd=%.1e gap=%.1e st=%s fixed=%d %.2fs" % (
            name, info["iterations"], int(e["iterations"]), info["objective"], float(e["objective"]),
            abs(info["objective"] - float(e["objective"])) / max(1, abs(float(e["objective"]))),
            info["rp"], info["rd"], info["gap"], info["status_name"], info["pivots_fixed"], time.time() - t0))


def stage_perf(m=4096, n=8192, steps=10):
    A, b, c = O.synthetic_lp(m, n)
    with ipm.IpmSolver(A, b, c) as sv:
        sv.init_state(0.0)
        sv.iterate(2)
        sv.init_state(0.0)
        st = sv.iterate(steps)
        print("perf %dx%d: %d steps %.3f ms/it  -> %.1f it/s (obj after %d its %.9e)" % (
            m, n, steps, st["solve_ms"] / steps, 1e3 * steps / st["solve_ms"], steps, st["objective"]))
        sv.init_state(0.0)
        sv.set_profiling(True)
        sv.iterate(steps)
        ph = sv.phase_ms()
        F = float(m) * m * n
        print("phases ms/it: form %.3f (%.1f TFLOP/s)  factor %.3f (%.1f TFLOP/s)  trisolve %.3f  other %.3f" % (
            ph["form"], F / ph["form"] / 1e9, ph["factor"], (m ** 3 / 3.0) / ph["factor"] / 1e9,
            ph["trisolve"], ph["other"]))
        sv.set_profiling(False)
        sv.init_state(0.0)
        st = sv.solve(tol=1e-8, max_iter=200)
        print("full solve: it=%d obj=%.12e status=%d  %.1f ms" % (st["iterations"], st["objective"], st["status"], st["solve_ms"]))


STAGES = dict(form=stage_form, chol=stage_chol, direction=stage_direction, solve=stage_solve, perf=stage_perf)

if __name__ == "__main__":
    want = sys.argv[1:] or ["form", "chol", "direction", "solve", "perf"]
    for nm in want:
        print("==== stage", nm, flush=True)
        try:
            STAGES[nm]()
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()
