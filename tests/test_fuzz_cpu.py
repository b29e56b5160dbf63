"""Differential fuzzing on CPU: oracle B against the reference's own binary (oracle A) on random
unorganised clouds and random parameters (tests/fuzz.py)."""
import numpy as np
import pytest

import oracles as O
from fuzz import case


# Cases in which ONE label depends on how the host's libm rounds asinf / acosf / atan2f (the
# reference's own answer is glibc-dependent on a decision boundary; this image's glibc 2.35 and include/urf_libm.h
# differ by 1 ulp there).  seed -> number of such labels; against the reference built with the shared libm (the test
# below covers every seed of this one) they are equal too.
GLIBC_DEPENDENT = {49: 1, 78: 2}   # (78: a tie-bearing cloud, x / y on a 1/16 m grid: many points share a vertical angle)


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("seed", range(100))
def test_oracle_b_equals_reference_on_random_input(seed):
    (x, y, z), p = case(1000 + seed, for_reference=True)
    la, ia, _, _ = O.run_a([(x, y, z)], p)
    lb, ib, _ = O.run_b(x, y, z, p)
    assert ia[0]["status"] == ib["status"]
    if seed in GLIBC_DEPENDENT:
        assert int((la[0] != (lb & O.MASK_NO_RING)).sum()) == GLIBC_DEPENDENT[seed]
        la2, ia2, _, _ = O.run_a([(x, y, z)], p, libm=True)
        assert np.array_equal(la2[0], lb & O.MASK_NO_RING)
        return
    assert np.array_equal(la[0], lb & O.MASK_NO_RING), "seed %d: %d labels differ" % (seed, int((la[0] != (lb & O.MASK_NO_RING)).sum()))
    if ib["status"] == 0:   # with < 30 ROI points the reference publishes nothing, not even its roi cloud
        for k in ("n_roi", "n_road", "n_curb", "n_ring10"):
            assert ia[0][k] == ib[k], k


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("seed", range(0, 140))
def test_oracle_b_equals_reference_with_shared_libm(seed):
    """Same comparison against the reference sources built with the product's definition of
    acosf / asinf / atan2f (oracle/shim/urf_libm_override.h): here equality has to hold for ANY
    input, because no difference between libm implementations is left."""
    (x, y, z), p = case(1000 + seed, for_reference=True)
    la, ia, _, _ = O.run_a([(x, y, z)], p, libm=True)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    assert ia[0]["status"] == ib["status"]
    assert np.array_equal(la[0], lb & O.MASK_NO_RING)
    if ib["status"] == 0:
        for key in ("road_order", "curb_order", "ring10_order"):
            assert np.array_equal(ia[0][key], st[key]), key


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("seed,pairs", [(201, 1500), (202, 40), (203, 1500)])
def test_oracle_b_equals_reference_with_nan_slopes(seed, pairs):
    """Identical points in a star sector: the slope between them is 0 / 0, which the reference counts and skips
    (star_shaped_search.cpp:131-132).  WHICH of two identical points carries the mark is decided by std::sort's order of
    equal ranges, which oracle B follows literally since r5 (oracle/urf_stdsort.h): exact equality."""
    from fuzz import cloud_with_identical_points
    p = O.cfg_params("cfg2")
    scan, involved = cloud_with_identical_points(seed, 64 * 2048, pairs)
    la, ia, _, _ = O.run_a([scan], p, libm=True)
    lb, ib, _ = O.run_b(*scan, p)
    assert np.array_equal(la[0], lb & O.MASK_NO_RING)
    for k in ("n_roi", "n_road", "n_curb", "n_ring10"):
        assert ia[0][k] == ib[k], k


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("scene,cfg,seed", [(3, "cfg2", 11), (3, "cfg2", 12), (4, "narrow", 13), (3, "default_roi", 14)])
def test_oracle_b_equals_reference_on_sensor_like_sweeps(scene, cfg, seed):
    """Sweeps as a sensor's driver delivers them (range noise, 2 mm range steps, drop-outs; ~10 000 exact planar-range ties
    per sweep, in every star sector): labels and published order equal the reference's.  With the stable order of r1-r4
    (ties by input index) 400-700 labels per sweep differed."""
    import urban_road_filter_amd as u
    x, y, z = u.synth_cloud(64, 2048, scene, seed)
    p = O.cfg_params(cfg)
    la, ia, _, _ = O.run_a([(x, y, z)], p, libm=True)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    assert np.array_equal(la[0], lb & O.MASK_NO_RING)
    for key in ("road_order", "curb_order", "ring10_order"):
        assert np.array_equal(ia[0][key], st[key]), key


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("n,dup", [(364, 0.0), (364, 0.1), (1024, 0.1), (4000, 0.05)])
def test_oracle_b_equals_reference_when_std_sort_reaches_its_depth_limit(n, dup):
    """One star sector whose ranges, in input order, are an adversarial sequence for libstdc++'s introsort (McIlroy's
    adversary run against the real std::sort, oracle/stdsort_ref.cpp): the reference's sort falls back to heap sort, whose
    order of equal ranges oracle B has to leave as well."""
    from fuzz import killer_sector_cloud
    import ctypes as C
    x, y, z = killer_sector_cloud(n, dup, seed=n)
    p = O.cfg_params("cfg2")
    p.interval = 1.0
    la, ia, _, _ = O.run_a([(x, y, z)], p, libm=True)
    hs = O.oracle_b().urf_oracle_std_sort_heap_sorts
    hs.restype = C.c_long
    before = hs()
    lb, ib, _ = O.run_b(x, y, z, p)
    assert hs() > before, "the sector did not reach the depth limit"
    assert ib["status"] == 0
    assert np.array_equal(la[0], lb & O.MASK_NO_RING)
