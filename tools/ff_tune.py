"""Evaluate work lists of ff_schedule.h (through ipm_debug_ff_schedule, host only) with the calibrated replay model of
tools/ff_replay.py.  Usage: python tools/ff_tune.py KEY=v1,v2 ...  (KEY = an IPM_FF_* environment knob of ff_build_schedule)"""
import ctypes as C
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interiorpointmethod_amd import _lib          # noqa: E402
from tools.ff_replay import replay, summary, Model, ModelRoles       # noqa: E402


def schedule(nblk=32, q=4, workers=224):
    lib = _lib.load()
    cap = 200000
    items = np.zeros(cap * 8, dtype=np.uint8)
    cnt = C.c_int32(0)
    ntile = nblk * (nblk + 1) // 2
    tile_items = np.zeros(ntile, dtype=np.int32)
    sim = (C.c_double * 2)()
    rc = lib.ipm_debug_ff_schedule(nblk, q, workers, items.ctypes.data_as(C.POINTER(C.c_ubyte)), cap, C.byref(cnt),
                                   tile_items.ctypes.data_as(C.POINTER(C.c_int32)), sim)
    assert rc == 0
    return items[:cnt.value * 8].reshape(-1, 8).copy(), (sim[0], sim[1])


if __name__ == "__main__":
    knobs = {}
    for a in sys.argv[1:]:
        k, v = a.split("=")
        knobs[k] = v.split(",")
    keys = list(knobs)
    for combo in itertools.product(*[knobs[k] for k in keys]):
        for k, v in zip(keys, combo):
            os.environ["IPM_FF_" + k] = v
        mode = int(os.environ.get("IPM_FF_CHAIN_MODE", "1"))
        os.environ["IPM_FF_CHAIN_MODE"] = str(mode)
        W = 251 if mode else 224
        items, sim = schedule(workers=W)
        try:
            r = replay(items, 32, W=W, M=ModelRoles if mode else Model, mode=mode)
            print(" ".join("%s=%s" % kv for kv in zip(keys, combo)), "| items %d | generator's own sim %.0f | replay: %s" % (len(items), sim[0], summary(r, W)))
        except RuntimeError as e:
            print(" ".join("%s=%s" % kv for kv in zip(keys, combo)), "| replay failed:", e)
