"""Sweeps in firing order with things MISSING -- what k_split's "organised tile with holes" path and k_ring's straddling quads see on
real data and the random unorganised clouds of tests/fuzz.py never produce: a synthetic 64-ring sweep (urf_synth_cloud, tie-free or
sensor-like) with random drop-outs (single points, whole firings, whole rings, azimuth ranges, runs inside a ring), points moved off
their ring or their sector, a random region of interest and random detector parameters.  GPU against oracle B (tests/test_gpu_organised.py,
tools/fuzz_organised_more.py)."""
import numpy as np

import urban_road_filter_amd as u


def case(seed):
    rng = np.random.default_rng(seed)
    cols = int(rng.choice([256, 512, 1024, 2048]))
    scene = int(rng.choice([1, 2, 3, 4]))
    x, y, z = u.synth_cloud(64, cols, scene, int(rng.integers(1, 1 << 30)))
    n = 64 * cols
    drop = np.zeros(n, bool)
    ring = np.arange(n) % 64
    col = np.arange(n) // 64
    kinds = rng.integers(0, 2, 7)
    if kinds[0]:
        drop |= rng.random(n) < float(rng.choice([0.002, 0.02, 0.2]))                       # single points
    if kinds[1]:
        drop |= np.isin(col, rng.integers(0, cols, int(rng.integers(1, 12))))               # whole firings
    if kinds[2]:
        drop |= np.isin(ring, rng.integers(0, 64, int(rng.integers(1, 6))))                 # whole rings
    if kinds[3]:
        a0 = int(rng.integers(0, cols))
        drop |= ((col - a0) % cols) < int(rng.integers(1, cols // 3))                       # an azimuth range
    if kinds[4]:
        for _ in range(int(rng.integers(1, 8))):                                            # a run inside one ring
            r, c0 = int(rng.integers(0, 64)), int(rng.integers(0, cols))
            drop |= (ring == r) & (((col - c0) % cols) < int(rng.integers(1, 200)))
    x[drop] = y[drop] = z[drop] = 0.0
    if kinds[5]:                                                                            # a few points off their ring / sector
        k = rng.integers(0, n, int(rng.integers(1, 20)))
        z[k] = (z[k] * rng.uniform(0.3, 1.7, len(k))).astype(np.float32)
        k = rng.integers(0, n, int(rng.integers(1, 20)))
        x[k], y[k] = y[k].copy(), x[k].copy()
    p = u.default_params()
    if rng.random() < 0.5:
        p = p.wide_roi()
    else:                                                                                   # a wedge / box that cuts rings and firings
        p.min_X, p.max_X = float(rng.choice([-200.0, 0.0, 3.0])), float(rng.choice([15.0, 30.0, 200.0]))
        p.min_Y, p.max_Y = float(rng.choice([-200.0, -10.0, -3.0])), float(rng.choice([2.0, 10.0, 200.0]))
    p.x_zero_method = int(rng.random() < 0.9)
    p.z_zero_method = int(rng.random() < 0.9)
    p.star_shaped_method = int(rng.random() < 0.85)
    p.blind_spots = int(rng.random() < 0.7)
    p.xDirection = int(rng.integers(0, 3))
    p.curbHeight = float(rng.choice([0.02, 0.05, 0.1]))
    p.curbPoints = int(rng.choice([5, 5, 5, 2, 9]))
    p.angleFilter1 = float(rng.choice([120.0, 150.0, 175.0]))
    p.angleFilter2 = float(rng.choice([100.0, 140.0, 170.0]))
    p.angleFilter3 = float(rng.choice([20.0, 30.0, 50.0]))
    p.starbeam_filter = int(rng.random() < 0.2)
    p.interval = float(rng.choice([0.18, 0.18, 0.1]))
    return (x, y, z), p
