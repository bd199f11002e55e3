"""Pins oracle/np_random.h against the installed numpy (the reference's RNG)."""
import ctypes as C

import numpy as np

from oracle_binding import load_oracle


def _replay(lib, seed, op, arg):
    out = np.zeros(len(op))
    lib.sso_t_replay(C.c_uint64(seed), len(op), op.ctypes.data_as(C.c_void_p), arg.ctypes.data_as(C.c_void_p),
                     out.ctypes.data_as(C.c_void_p))
    return out


def test_seed_sequence_and_pcg64_state():
    lib = load_oracle()
    for seed in list(range(64)) + [1234, 2**32 - 1, 2**32, 2**40 + 12345, 2**63 + 5]:
        out = (C.c_uint64 * 4)()
        lib.sso_t_seed_state(C.c_uint64(seed), out)
        st = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed))).bit_generator.state["state"]
        exp = [st["state"] >> 64, st["state"] & (2**64 - 1), st["inc"] >> 64, st["inc"] & (2**64 - 1)]
        assert list(out) == exp, seed


def test_mixed_stream_matches_numpy():
    """random() / integers(n) / exponential interleaved: the 32-bit buffer must persist across
    random() calls and integers(1) must consume nothing."""
    lib = load_oracle()
    rs = np.random.default_rng(99)
    for seed in range(60):
        n = 300
        op = rs.integers(0, 3, size=n).astype(np.int32)
        arg = rs.choice([1, 2, 3, 5, 7, 22, 100, 220, 1000, 2**31, 2**32 - 1], size=n).astype(np.uint32)
        got = _replay(lib, seed, op, arg)
        g = np.random.default_rng(seed)
        exp = np.zeros(n)
        for i in range(n):
            if op[i] == 0:
                exp[i] = g.random()
            elif op[i] == 1:
                exp[i] = g.integers(int(arg[i]))
            else:
                exp[i] = g.exponential(1.0)
        assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), seed


def test_choice_equals_integers():
    g1, g2 = np.random.default_rng(5), np.random.default_rng(5)
    for n in (1, 2, 7, 13, 220):
        lst = list(range(100, 100 + n))
        for _ in range(50):
            assert g1.choice(lst) == lst[g2.integers(n)]


def test_exponential_ziggurat_bit_exact_including_tail():
    lib = load_oracle()
    n = 1_500_000
    op = np.full(n, 2, dtype=np.int32)
    arg = np.zeros(n, dtype=np.uint32)
    got = _replay(lib, 1, op, arg)
    exp = np.random.default_rng(1).exponential(1.0, size=n)
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
    assert (got > 7.69711747013104972).sum() > 100  # the log1p tail was exercised


def test_log1p_matches_libm_on_domain():
    lib = load_oracle()
    libm = C.CDLL("libm.so.6")
    libm.log1p.restype = C.c_double
    libm.log1p.argtypes = [C.c_double]
    x = -np.random.default_rng(5).random(100_000)
    x[:4] = [-0.0, -1e-300, -2.0**-30, -0.999999999999]
    y = np.zeros_like(x)
    lib.sso_t_log1p(len(x), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    ref = np.array([libm.log1p(v) for v in x])
    assert np.array_equal(y.view(np.uint64), ref.view(np.uint64))


def test_exp_within_one_ulp():
    lib = load_oracle()
    x = -np.random.default_rng(6).random(200_000) * 45
    y = np.zeros_like(x)
    lib.sso_t_exp(len(x), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
    d = np.abs(y.view(np.int64) - np.exp(x).view(np.int64))
    assert d.max() <= 1
