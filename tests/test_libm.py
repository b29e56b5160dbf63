"""include/urf_libm.h (the shared acosf/asinf/atan2f) through oracle B's exports:
correctly rounded w.r.t. a binary64 evaluation and within 1 ulp of the host libm."""
import ctypes
import os

import numpy as np
import pytest

import oracles as O

ROOT = O.ROOT


def _vec(fn, *args):
    L = O.oracle_b()
    f = getattr(L, fn)
    return np.array([f(*[ctypes.c_float(float(a)) for a in t]) for t in zip(*args)], np.float32)


_M = ctypes.CDLL("libm.so.6")
for _f in ("asinf", "acosf"):
    getattr(_M, _f).argtypes = [ctypes.c_float]
    getattr(_M, _f).restype = ctypes.c_float
_M.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
_M.atan2f.restype = ctypes.c_float


def _libm(fn, *args):
    """glibc's float functions -- what the reference itself calls on this host."""
    f = getattr(_M, fn)
    return np.array([f(*[float(a) for a in t]) for t in zip(*args)], np.float32)


def _ulps(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_asin_acos_match_rounded_double():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-1, 1, 20000), [0, 1, -1, 0.5, -0.5, 1e-20, 0.99999994, -0.99999994]]).astype(np.float32)
    a = _vec("urf_oracle_asinf", x)
    c = _vec("urf_oracle_acosf", x)
    assert np.array_equal(a, np.arcsin(x.astype(np.float64)).astype(np.float32))
    assert np.array_equal(c, np.arccos(x.astype(np.float64)).astype(np.float32))
    # within 1 ulp of the host's float libm (what the reference calls)
    assert _ulps(a, _libm('asinf', x)).max() <= 1
    assert _ulps(c, _libm('acosf', x)).max() <= 1


def test_atan2_matches_rounded_double():
    rng = np.random.default_rng(1)
    y = rng.uniform(-100, 100, 20000).astype(np.float32)
    x = rng.uniform(-100, 100, 20000).astype(np.float32)
    r = _vec("urf_oracle_atan2f", y, x)
    assert np.array_equal(r, np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32))
    assert _ulps(r, _libm('atan2f', y, x)).max() <= 1


def test_special_values():
    L = O.oracle_b()
    assert L.urf_oracle_acosf(1.0) == 0.0
    assert L.urf_oracle_asinf(0.0) == 0.0
    assert np.isnan(L.urf_oracle_acosf(1.5)) and np.isnan(L.urf_oracle_asinf(float("nan")))
    pi = np.float32(np.pi)
    assert np.float32(L.urf_oracle_acosf(-1.0)) == pi
    assert np.float32(L.urf_oracle_atan2f(0.0, -1.0)) == pi
    assert np.float32(L.urf_oracle_atan2f(-0.0, -1.0)) == -pi
    assert L.urf_oracle_atan2f(0.0, 0.0) == 0.0
    assert np.float32(L.urf_oracle_atan2f(1.0, 0.0)) == np.float32(np.pi / 2)


def test_ring_threshold_cotangent():
    """k_ring_table turns every ring-table entry's window into thresholds on u = cot(vertical angle) with a
    binary64 cotangent built from Taylor series (urf_device.hpp: urf_cot_deg); the library exports the host
    evaluation of the same source.  Needed: 1e-7 relative; measured here against numpy."""
    import ctypes as C
    import urban_road_filter_amd as u
    lib = u.api.lib()
    lib.urf_ring_threshold_cot.restype = C.c_double
    lib.urf_ring_threshold_cot.argtypes = [C.c_double]
    deg = np.concatenate([np.linspace(1.0, 179.0, 20001), 90.0 + np.array([-1e-9, 0.0, 1e-9]), [14.0362, 165.9638]])
    got = np.array([lib.urf_ring_threshold_cot(float(d)) for d in deg])
    want = 1.0 / np.tan(np.deg2rad(deg))
    # (near 90 degrees, where the cotangent passes through zero, the comparison is absolute: numpy's own
    # deg2rad rounds the argument there)
    excess = np.abs(got - want) - (1e-13 * np.abs(want) + 1e-15)
    assert excess.max() <= 0.0, (excess.max(), deg[np.argmax(excess)])
    # clamped outside [1, 179] degrees: beyond every u the fast path accepts (|u| <= 4)
    assert lib.urf_ring_threshold_cot(0.2) == lib.urf_ring_threshold_cot(1.0) > 57.0
    assert lib.urf_ring_threshold_cot(179.9) == lib.urf_ring_threshold_cot(179.0) < -57.0
    # strictly decreasing: what turns "the angle lies in a window" into "u lies between two thresholds"
    assert np.all(np.diff(got[:20001]) < 0)



def test_angle_filter_as_a_threshold_on_the_cosine(tmp_path):
    """x_zero / z_zero test "alpha <= angleFilter", alpha = acosf(b) in degrees; the kernels test "b >= T" with T found by bisection
    when the parameters are set (urf_api.hip: urf_angle_threshold).  Valid because alpha falls as b grows: tools/check_acos_threshold.c
    checks that, and the bisection, against include/urf_libm.h (full run: every float of [-1, 1]; here its quick mode)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "check_acos")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "check_acos_threshold.c"), "-o", exe, "-lm"])
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "alpha(b) rises somewhere: 0 of" in r.stdout
