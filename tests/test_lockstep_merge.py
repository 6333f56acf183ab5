"""The schedule merge of the lockstep batch (csrc/lockstep_merge.h) on the CPU, through ipm_debug_ls_merge: every LP's launch
order is preserved, every step holds one kernel type and at most max_group LPs, and programs that follow one template (the shape of
an interior-point iteration: main.py:780-807 of the reference) merge into about the LONGEST program, where the leader rule the
alignment replaced gives about the sum of two out-of-phase programs."""
import ctypes as C

import numpy as np
import pytest

from interiorpointmethod_amd import _lib

P, G, U, F, V, T = 9, 10, 11, 6, 22, 16      # potrf, panel, update, formation, gemv, substitution step (lockstep.h LsType values)


def merge(progs, max_group=64, aligned=1):
    lib = _lib.load()
    off = np.zeros(len(progs) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(p) for p in progs])
    flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int32) for p in progs]) if progs else np.zeros(0, np.int32))
    total = int(off[-1])
    cap = 2 * total + 1
    steps = np.zeros(2 * cap, dtype=np.int32)
    members = np.zeros(2 * max(total, 1), dtype=np.int32)
    i32p = C.POINTER(C.c_int32)
    n = lib.ipm_debug_ls_merge(len(progs), off.ctypes.data_as(i32p), flat.ctypes.data_as(i32p), max_group, aligned,
                               steps.ctypes.data_as(i32p), cap, members.ctypes.data_as(i32p))
    assert n >= 0
    out, w = [], 0
    for s in range(n):
        t, cnt = int(steps[2 * s]), int(steps[2 * s + 1])
        out.append((t, [(int(members[2 * (w + k)]), int(members[2 * (w + k) + 1])) for k in range(cnt)]))
        w += cnt
    assert w == total
    return out


def check(progs, plan, max_group=64):
    pos = [0] * len(progs)
    for t, mem in plan:
        assert 1 <= len(mem) <= max_group
        assert len({i for i, _ in mem}) == len(mem)              # an LP takes part in a step at most once
        for i, j in mem:
            assert j == pos[i] and progs[i][j] == t              # in order, nothing skipped, one kernel type per step
            pos[i] += 1
    assert pos == [len(p) for p in progs]


def iteration(nb, pre=(0, 1, 2, 3), form=(F,), post=(V, T, T, V, 18, 19, 20, T, T, 18, 21)):
    prog = list(pre) + list(form)
    for k in range(nb):
        prog += [P] if k == nb - 1 else [P, G, U]
    return prog + list(post)


def test_same_template_merges_to_the_longest_program():
    progs = [iteration(nb) for nb in (16, 5, 9, 16, 2, 12)]
    plan = merge(progs)
    check(progs, plan)
    assert len(plan) == len(progs[0])
    # the leader rule: also fine here (the programs are in phase)
    check(progs, merge(progs, aligned=0))


def test_out_of_phase_programs():
    # the same loop behind formation sections of different length: the leader rule alternates between the two LPs
    a = iteration(16, form=(4, 5))
    b = iteration(16, form=(F,))
    c = iteration(15, form=(7,), post=(V, 23, 24, T, V, 18, 19, 20, T, 23, 24, 18, 21))
    progs = [a, b, c]
    plan = merge(progs)
    check(progs, plan)
    old = merge(progs, aligned=0)
    check(progs, old)
    assert len(plan) <= len(a) + 8
    assert len(plan) < len(old)


def test_group_limit_and_edge_cases():
    progs = [iteration(3) for _ in range(70)]
    plan = merge(progs, max_group=64)
    check(progs, plan, 64)
    assert len(plan) == 2 * len(progs[0])
    assert merge([]) == []
    one = [iteration(4)]
    assert [t for t, _ in merge(one)] == one[0]
    check([[], iteration(2)], merge([[], iteration(2)]))


def test_random_programs_keep_every_order():
    rng = np.random.default_rng(0)
    for _ in range(20):
        progs = [list(rng.integers(0, 5, size=int(rng.integers(0, 40)))) for _ in range(int(rng.integers(1, 12)))]
        for aligned in (0, 1):
            check(progs, merge(progs, max_group=4, aligned=aligned), 4)
        # never longer than running the programs one after the other
        assert len(merge(progs, max_group=64)) <= max(1, sum(len(p) for p in progs))
