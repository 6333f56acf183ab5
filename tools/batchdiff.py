#!/usr/bin/env python3
"""Diagnostic: the 73-LP Netlib suite through batch.run_batch with two LPs in flight, one at a time, and two in flight again;
prints every LP whose (status, iterations, objective) record differs between the runs (expected: none -- the sync mechanism
of concurrent handles must not change arithmetic)."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from interiorpointmethod_amd import batch
from interiorpointmethod_amd.matio import load_npz_problem
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
names, probs = [], []
for f in sorted(glob.glob(os.path.join(R, "tests", "golden", "netlib", "*.npz"))):
    A, b, c, cTlb, valid = load_npz_problem(f)
    if valid:
        names.append(os.path.basename(f)[:-4]); probs.append((A, b, c))
par, _ = batch.run_batch(probs, tol=1e-8, max_iter=300, workers=2)
seq, _ = batch.run_batch(probs, tol=1e-8, max_iter=300, workers=1)
par2, _ = batch.run_batch(probs, tol=1e-8, max_iter=300, workers=2)
for i in range(len(names)):
    a, b2, c2 = seq[i], par[i], par2[i]
    eq = lambda u, v: (u[1] == v[1] and u[2] == v[2] and (u[3] == v[3] or (np.isnan(u[3]) and np.isnan(v[3]))))
    if not eq(a, b2) or not eq(b2, c2):
        print(names[i], probs[i][0].shape, "seq", a[1:4], "par", b2[1:4], "par2", c2[1:4])
print("done")
