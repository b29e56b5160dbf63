"""oracle B's restatement of libstdc++'s std::sort (oracle/urf_stdsort.h) against the real thing
(oracle/stdsort_ref.cpp -> libstdsort_ref.so): the order of EQUAL planar ranges inside a star
sector decides labels (star_shaped_search.cpp:109, 123-149), std::sort is not stable, and the
reference's answer is whatever this algorithm leaves -- so the restatement has to leave the same.
Every sequence is compared record by record (the ids tell equal ranges apart)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracles as O

REF = os.path.join(O.ORACLE_DIR, "libstdsort_ref.so")


def _libs():
    O.ensure_built()
    ref = C.CDLL(REF)
    b = O.oracle_b()
    for f in (ref.urf_ref_std_sort, b.urf_oracle_std_sort):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        f.restype = None
    return ref.urf_ref_std_sort, b.urf_oracle_std_sort


def _both(r):
    ref, mine = _libs()
    r = np.ascontiguousarray(r, np.float32)
    out = []
    for f in (ref, mine):
        rr, ii = r.copy(), np.arange(len(r), dtype=np.int32)
        f(rr.ctypes.data, ii.ctypes.data, len(r))
        assert np.all(rr[1:] >= rr[:-1])
        out.append(ii)
    return out


def killer(n):
    """An input on which THIS std::sort reaches its depth limit 2 * floor(log2 n), so that the heap sort of
    stl_algo.h:1909-1918 runs (McIlroy's adversary against the real std::sort, oracle/stdsort_ref.cpp)."""
    O.ensure_built()
    ref = C.CDLL(REF)
    ref.urf_ref_killer.argtypes = [C.c_void_p, C.c_int]
    a = np.zeros(n, np.float32)
    ref.urf_ref_killer(a.ctypes.data, n)
    return a


@pytest.mark.parametrize("n", [0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 100, 364, 365, 1000, 1456, 5000])
def test_sizes_with_ties(n):
    rng = np.random.default_rng(n)
    for alphabet in (1, 2, 3, 7, 40, 10 ** 6):
        r = rng.integers(0, alphabet, n).astype(np.float32)
        a, b = _both(r)
        assert np.array_equal(a, b), (n, alphabet)


def test_shapes():
    rng = np.random.default_rng(5)
    for n in (17, 40, 364, 2048, 4097):
        base = np.sort(rng.integers(0, max(2, n // 3), n)).astype(np.float32)
        for r in (base, base[::-1], np.concatenate([base[::2], base[1::2][::-1]]), np.roll(base, n // 3),
                  np.where(rng.random(n) < 0.1, rng.integers(0, 50, n), base).astype(np.float32)):
            a, b = _both(r)
            assert np.array_equal(a, b), n


def test_heap_sort_fallback_is_reached_and_followed():
    hs = O.oracle_b().urf_oracle_std_sort_heap_sorts
    hs.restype = C.c_long
    for n in (64, 200, 364, 1024, 4000):
        r = killer(n)
        before = hs()
        a, b = _both(r)
        assert np.array_equal(a, b), n
        assert hs() > before, "the killer sequence did not reach the depth limit" 
        # ... with ties on top (quantised): the fallback's order of equal keys is the heap's
        a, b = _both(np.floor(r / 3))
        assert np.array_equal(a, b), n


def test_sensor_like_sectors():
    """ranges of a sector of a quantised sweep: ~6 firings x 64 rings, range noise, 2 mm steps"""
    rng = np.random.default_rng(11)
    for _ in range(300):
        rings = 1.8 / np.tan(np.deg2rad(np.linspace(2.0, 24.8, 64)))
        r = np.repeat(rings, 6) + 0.01 * rng.standard_normal(384)
        r = (np.round(r / 0.002) * 0.002).astype(np.float32)
        r = r[rng.permutation(384)] if rng.random() < 0.3 else r.reshape(64, 6).T.ravel()
        a, b = _both(r[: int(rng.integers(200, 385))])
        assert np.array_equal(a, b)


def test_many_random():
    rng = np.random.default_rng(2)
    for _ in range(3000):
        n = int(rng.integers(0, 600))
        r = rng.integers(0, int(rng.choice([2, 5, 30, 300, 10 ** 5])), n).astype(np.float32)
        if rng.random() < 0.3:
            r = np.sort(r)
        a, b = _both(r)
        assert np.array_equal(a, b)
