#!/usr/bin/env python3
"""Diagnostic: phase timing (s_memtime) inside potrf_diag_kernel for one 128x128 block."""
import ctypes as C, os, sys
import numpy as np
os.environ["IPM_POTRF_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import interiorpointmethod_amd as ipm
m = 128
rng = np.random.default_rng(0)
M = rng.standard_normal((m, m + 10)); B = M @ M.T + 0.1 * np.eye(m); rhs = rng.standard_normal(m)
with ipm.IpmSolver(np.eye(m, 1), np.zeros(m), np.zeros(1)) as sv:
    for _ in range(3):
        z, nfix = sv.solve_linear(B, rhs)
    buf = (C.c_longlong * 512)()
    sv._lib.ipm_debug_get_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    assert sv._lib.ipm_debug_get_stamps(sv._h, buf) == 0
st = np.array(buf[:], dtype=np.int64).reshape(8, 64)
t0 = st[:, 0].min()
print("resid", np.linalg.norm(B @ z.ravel() - rhs) / np.linalg.norm(rhs))
print("cycles since kernel start, per wave: load_done, factor0_done(w0), after_sync")
print(st[:, 1] - t0, st[:, 2] - t0, st[:, 3] - t0)
for jb in range(8):
    a, b, c, d = (st[:, 4 + 4 * jb + i] - t0 for i in range(4))
    print("jb%d: P2 end per wave %s | sync %d | P3 end per wave %s | sync %d" % (jb, a, b.max(), c, d.max()))
print("row7 assembly end", st[:, 38] - t0)
print("end", st[:, 40] - t0, " total cycles", (st[:, 40] - t0).max())
