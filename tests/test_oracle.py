"""Pins oracle B (oracle/urf_oracle.c, the C restatement):
  - against the golden label vectors in tests/golden/ (produced by oracle A = the reference's own
    unmodified sources, tests/golden/make_golden.py), on every machine;
  - against oracle A itself, live, where its binary exists (built from /root/reference).
Labels must match bit for bit (the RING bit 0x08 is not observable from the reference)."""
import glob
import hashlib
import os

import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u
from golden.make_golden import CASES, case_params, cloud_sha

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_golden_files_present():
    assert len(glob.glob(os.path.join(GOLD, "*.npz"))) == len(CASES)


@pytest.mark.parametrize("name,cfg,seed,tweak", CASES, ids=[c[0] for c in CASES])
def test_oracle_b_equals_golden(name, cfg, seed, tweak):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    p = case_params(cfg, tweak)
    assert bytes(p) == g["params"].tobytes()
    x, y, z = O.case_cloud(cfg, seed, g)
    assert cloud_sha(x, y, z) == str(g["cloud_sha"]), "synthetic generator drifted"
    lb, ib, _ = O.run_b(x, y, z, p)
    assert np.array_equal(lb & O.MASK_NO_RING, g["labels"])
    for k in ("status", "n_roi", "n_road", "n_curb", "n_ring10"):
        assert ib[k] == int(g["info_" + k]), k
    # internal consistency of the label byte
    road = (lb & 3) == 1
    curb = (lb & 3) == 2
    assert ((lb & u.FLAG_RING) != 0)[road | curb].all() and ((lb & u.FLAG_ROI) != 0)[(lb & u.FLAG_RING) != 0].all()
    assert road.sum() == ib["n_road"] and curb.sum() == ib["n_curb"]


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("cfg,seeds,tweak", [
    ("cfg1", [2, 3], {}),
    ("cfg2", [4, 5], {}),
    ("narrow", [2], {"xDirection": 1, "curbPoints": 7}),
    ("default_roi", [2], {"z_zero_method": 0}),
    ("cfg2", [6], {"starbeam_filter": 1, "beamZone": 20.0, "kdev_param": 0.8}),
])
def test_oracle_b_equals_oracle_a_live(cfg, seeds, tweak):
    p = case_params(cfg, tweak)
    scans = [O.cfg_cloud(cfg, s) for s in seeds]
    la, ia, _, _ = O.run_a(scans, p)
    for k in range(len(seeds)):
        lb, ib, st = O.run_b(*scans[k], p, debug=True)
        assert np.array_equal(la[k], lb & O.MASK_NO_RING), (cfg, seeds[k])
        assert ia[k]["n_road"] == ib["n_road"] and ia[k]["n_curb"] == ib["n_curb"] and ia[k]["n_roi"] == ib["n_roi"]
        # the clouds in the order the reference published them (ring-major, azimuth ascending)
        for key in ("road_order", "curb_order", "ring10_order"):
            assert np.array_equal(ia[k][key], st[key]), key


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("log2_scale", [0, -30, 30, -62])
@pytest.mark.parametrize("tweak", [{}, {"starbeam_filter": 1, "xDirection": 1}, {"channels": 20}, {"curbPoints": 9}])
def test_boundary_cloud_live(log2_scale, tweak):
    """Points ON the ring / sector / integer-degree decisions (O.boundary_cloud), at four scales: the
    restatement must equal the reference sources exactly when both use the same definition of
    acosf / asinf / atan2f.  (With the host's glibc the reference's own answer depends on the glibc
    release there: 2.35 labels 3 of the 3472 points differently, asinf(0.8660254f) being 1 ulp high.)"""
    sc = 2.0 ** log2_scale
    x, y, z = O.boundary_cloud(sc)
    p = u.default_params()
    for k, v in tweak.items():
        setattr(p, k, v)
    p.min_X, p.max_X, p.min_Y, p.max_Y, p.min_Z, p.max_Z = -60 * sc, 60 * sc, -60 * sc, 60 * sc, -3 * sc, -1 * sc
    p.channels = tweak.get("channels", 32)
    la, ia, _, _ = O.run_a([(x, y, z)], p, libm=True)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    assert np.array_equal(la[0], lb & O.MASK_NO_RING)
    for key in ("road_order", "curb_order", "ring10_order"):
        assert np.array_equal(ia[0][key], st[key]), key


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
def test_too_few_points_live():
    """< 30 ROI points: the reference publishes nothing (lidar_segmentation.cpp:124-126)."""
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("cfg2", 1)
    x, y, z = x[:4096].copy(), y[:4096].copy(), z[:4096].copy()
    z[29:] = 5.0   # outside the z ROI
    la, ia, _, _ = O.run_a([(x, y, z)], p)
    lb, ib, _ = O.run_b(x, y, z, p)
    assert ia[0]["status"] == 1 == ib["status"] and not la[0].any() and not lb.any()
    z[29] = -1.8
    la, ia, _, _ = O.run_a([(x, y, z)], p)
    lb, ib, _ = O.run_b(x, y, z, p)
    assert ia[0]["status"] == 0 == ib["status"] and np.array_equal(la[0], lb & O.MASK_NO_RING) and ib["n_roi"] == 30


def test_storage_order_invariance():
    """Ring-major and firing-order storage of the same sweep give the same label per point: the
    within-ring order (what x_zero/z_zero see) is the same in both (hard part 3)."""
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("cfg2", 3)
    perm = np.arange(64 * 2048).reshape(2048, 64).T.reshape(-1)   # ring-major order
    l1, _, _ = O.run_b(x, y, z, p)
    l2, _, _ = O.run_b(x[perm], y[perm], z[perm], p)
    assert np.array_equal(l1[perm], l2)


def test_nan_and_zero_points_are_dropped():
    p = O.cfg_params("cfg2")
    x, y, z = [a.copy() for a in O.cfg_cloud("cfg2", 1)]
    x[100] = np.nan
    x[200] = y[200] = z[200] = 0.0
    x[300], y[300], z[300] = 1.0, 1.0, -2.0   # x+y+z == 0
    lb, _, _ = O.run_b(x, y, z, p)
    assert lb[100] == 0 and lb[200] == 0 and lb[300] == 0


def test_sha_of_goldens_listed():
    """Human-checkable digest list, also printed by make_golden.py."""
    for name, *_ in CASES:
        g = np.load(os.path.join(GOLD, name + ".npz"))
        assert len(hashlib.sha256(g["labels"].tobytes()).hexdigest()) == 64
