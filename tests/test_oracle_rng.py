"""RNG pins (CPU): the oracle's XORWOW against rocRAND, the published constants, and the
product's host-side generator against the oracle's.

The reference holds no RNG vector; cuRAND's seed salts stay "parity
unpinned".  What is pinned here: recurrence + Weyl step + 2^67 sequence jump (rocRAND),
curand_uniform's range, jump composition, and oracle == product for every state word.
"""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import oraclelib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u32p = C.POINTER(C.c_uint32)


def _p(a):
    return a.ctypes.data_as(u32p)


def test_marsaglia_constants_seed_zero_salts():
    """curand_init with t0 = t1 = 0 is Marsaglia's xorwow start vector; check the salts cancel:
    seed = 0xf7dcefdd_aad26b49 makes s0 = s1 = 0."""
    seed = (0xf7dcefdd << 32) | 0xaad26b49
    st = oraclelib.rng_init(seed, 1)[0]
    assert list(st) == [6615241, 123456789, 362436069, 521288629, 88675123, 5783321]


def test_first_draws_follow_the_recurrence():
    st = oraclelib.rng_init(1024, 1)[0].copy()
    d, v = int(st[0]), [int(x) for x in st[1:]]
    L = oraclelib.lib()
    for _ in range(100):
        t = (v[0] ^ (v[0] >> 2)) & 0xffffffff
        v = v[1:] + [((v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1))) & 0xffffffff]
        d = (d + 362437) & 0xffffffff
        assert L.orc_rng_next(_p(st)) == (v[4] + d) & 0xffffffff


def test_uniform_range_endpoints():
    """curand_uniform maps x=0 -> 2^-33 (>0) and x=0xffffffff -> 1.0f: range (0, 1]."""
    f = np.float32
    lo = f(f(0) * f(2.3283064e-10) + f(2.3283064e-10) / f(2))
    hi = f(f(np.uint32(0xffffffff)) * f(2.3283064e-10) + f(2.3283064e-10) / f(2))
    assert lo > 0 and abs(float(lo) - 2.0 ** -33) < 1e-20
    assert hi == f(1.0)
    assert f(2.3283064e-10) == f(2.0 ** -32)  # exact power of two: mul+add == fma
    st = oraclelib.rng_init(7, 1)[0].copy()
    L = oraclelib.lib()
    xs = np.array([L.orc_rng_uniform(_p(st)) for _ in range(20000)], dtype=np.float32)
    assert xs.min() > 0 and xs.max() <= 1.0
    assert abs(xs.mean() - 0.5) < 0.01


def test_random_float_is_mul_add():
    st1 = oraclelib.rng_init(3, 1)[0].copy()
    st2 = st1.copy()
    L = oraclelib.lib()
    for mn, mx in [(-1, 1), (0, 0.9), (0, 0.5), (2, 5)]:
        t = np.float32(L.orc_rng_uniform(_p(st1)))
        want = np.float32(np.float32(t * np.float32(np.float32(mx) - np.float32(mn))) + np.float32(mn))
        got = np.float32(L.orc_random_float(C.c_float(mn), C.c_float(mx), _p(st2)))
        assert got == want


def test_jump_composition_and_d_unchanged():
    """subsequence a+b == jump(a) then jump(b); d is untouched by sequence jumps."""
    L = oraclelib.lib()
    base = oraclelib.rng_init(99, 1)[0]
    for a, b in [(1, 2), (5, 8), (1000, 24), (65536, 3)]:
        sa = oraclelib.rng_init(99, 1, first=a)[0]
        sab = oraclelib.rng_init(99, 1, first=a + b)[0]
        assert sa[0] == base[0] == sab[0]
        v = sa[1:].copy()
        k = 0
        bb = b
        while bb:
            if bb & 1:
                L.orc_rng_jump_pow2(_p(v), k)
            bb >>= 1
            k += 1
        assert np.array_equal(v, sab[1:])


def test_jump_is_not_a_small_power():
    """sanity: subsequence 1 differs from the first 10^4 states of subsequence 0."""
    L = oraclelib.lib()
    s1 = oraclelib.rng_init(5, 1, first=1)[0][1:]
    v = oraclelib.rng_init(5, 1)[0][1:].copy()
    for _ in range(10000):
        L.orc_rng_step_v(_p(v))
        assert not np.array_equal(v, s1)


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc (for rocRAND's host headers) not available")
def test_recurrence_and_2p67_jump_match_rocrand():
    """oracle/rocrand_xcheck.cc: 60 (seed, subsequence) pairs x 32 draws equal rocRAND's XORWOW
    when the oracle is given rocRAND's salts (host-only program, no GPU)."""
    odir = os.path.join(ROOT, "oracle")
    subprocess.run(["make", "-C", odir, "_build/rocrand_xcheck"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(odir, "_build", "rocrand_xcheck")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK")


def test_product_host_rng_equals_oracle():
    import rtmi
    L = rtmi.lib()
    for seed in [0, 1024, 10086, 0x123456789abcdef0]:
        for sub in [0, 1, 2, 63, 64, 1000, 65535, 1048575, 1048581, 16777215, (1 << 32) + 7]:
            mine = np.zeros(6, dtype=np.uint32)
            assert L.rtmi_rng_host_state(C.c_uint64(seed), C.c_uint64(sub), _p(mine)) == 0
            want = oraclelib.rng_init(seed, 1, first=sub)[0]
            assert np.array_equal(mine, want), (seed, sub)
    a = oraclelib.rng_init(10086, 1)[0].copy()
    b = a.copy()
    OL = oraclelib.lib()
    for mn, mx in [(0, 1), (-1, 1), (0, 0.9)] * 50:
        x = OL.orc_random_float(C.c_float(mn), C.c_float(mx), _p(a))
        y = L.rtmi_rng_host_random_float(C.c_float(mn), C.c_float(mx), _p(b))
        assert np.float32(x) == np.float32(y)
    assert np.array_equal(a, b)
