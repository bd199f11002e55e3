"""Differential test of oracle/pyset.h against the live CPython `set`: pool
add/remove churn, `.copy()`, `set(generator)` + drain by `pop()`, `set(list + list)`, iteration."""
import ctypes as C
import random
import sys

import pytest

from oracle_binding import load_oracle


@pytest.mark.skipif(sys.version_info[:2] != (3, 10), reason="model is of CPython 3.10's set")
def test_pyset_model_matches_cpython():
    lib = load_oracle()
    buf = (C.c_int32 * 8192)()

    def clist(h):
        n = lib.pst_list(h, buf)
        return list(buf[:n])

    rnd = random.Random(1)
    ops = 0
    for _ in range(600):
        K = rnd.choice([5, 10, 50, 64, 200, 1000])
        for h in range(4):
            lib.pst_clear(h)
        py = [set(), set()]
        if rnd.random() < 0.5:
            n0 = rnd.randrange(0, K + 1)
            py[0] = set(range(n0))
            for i in range(n0):
                lib.pst_add(0, i)
        for _ in range(rnd.randrange(10, 300)):
            h = rnd.randrange(2)
            r = rnd.random()
            if r < 0.35:
                k = rnd.randrange(K)
                py[h].add(k)
                lib.pst_add(h, k)
            elif r < 0.6:
                if py[h]:
                    k = rnd.choice(list(py[h]))
                    py[h].remove(k)
                    assert lib.pst_remove(h, k) == 1
            elif r < 0.7:
                if py[h]:
                    assert py[h].pop() == lib.pst_pop(h)
            elif r < 0.8:
                c = py[h].copy()
                lib.pst_copy(2, h)
                assert list(c) == clist(2)
                m = rnd.randrange(1, 4)
                f = set(x for x in c if x % m != 0)
                lib.pst_clear(3)
                for x in clist(2):
                    if x % m != 0:
                        lib.pst_add(3, x)
                assert list(f) == clist(3)
                while f and rnd.random() > 0.1:
                    assert f.pop() == lib.pst_pop(3)
            elif r < 0.9:
                lst = [rnd.randrange(K) for _ in range(rnd.randrange(0, 60))]
                lib.pst_clear(3)
                for x in lst:
                    lib.pst_add(3, x)
                assert list(set(lst)) == clist(3)
            assert list(py[h]) == clist(h)
            ops += 1
    assert ops > 50_000
