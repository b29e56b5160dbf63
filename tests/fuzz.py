"""Random unorganised clouds and random parameter sets for differential testing
(oracle B vs the reference binary on CPU, HIP path vs oracle B on the GPU).

The clouds deliberately violate everything an organised sweep guarantees: arbitrary point
order, uneven rings, empty / crowded sectors, points outside the ROI, NaNs, few points.
What they avoid is only what the REFERENCE ITSELF leaves undefined: the
sector-360 band (null dereference).  Points with x == y == 0 -- a NaN azimuth inside the reference's
Lomuto quicksort, which handles it deterministically -- are part of a third of the cases (axis_points);
exact planar-range ties inside a star sector -- ordered by libstdc++'s std::sort, deterministic as well
and followed since r5 (oracle/urf_stdsort.h, k_star_ties) -- of two fifths of them (x / y snapped to a
grid: many points share their (x, y) and differ in z, the kind of tie that decides labels)."""
import numpy as np

import urban_road_filter_amd as u


def random_params(rng, for_reference=False):
    p = u.default_params()
    p.x_zero_method = int(rng.random() < 0.8)
    p.z_zero_method = int(rng.random() < 0.8)
    p.star_shaped_method = int(rng.random() < 0.8)
    p.blind_spots = int(rng.random() < 0.7)
    p.xDirection = int(rng.integers(0, 3))
    p.interval = float(rng.choice([0.05, 0.18, 0.5, 1.5]))
    p.curbHeight = float(rng.choice([0.01, 0.05, 0.2]))
    p.curbPoints = int(rng.choice([1, 2, 5, 9, 30]))
    p.beamZone = float(rng.choice([10.0, 30.0, 45.5, 100.0]))
    half = float(rng.choice([15.0, 60.0, 200.0]))
    p.min_X, p.max_X, p.min_Y, p.max_Y = -half, half, -half * 0.8, half
    p.min_Z, p.max_Z = -3.0, float(rng.choice([-1.0, 0.5]))
    p.angleFilter1 = float(rng.choice([120.0, 150.0, 175.0]))
    p.angleFilter2 = float(rng.choice([100.0, 140.0, 170.0]))
    p.angleFilter3 = float(rng.choice([5.0, 50.0, 80.0]))
    p.kdev_param = float(rng.choice([0.5, 1.225, 5.0]))
    p.kdist_param = float(rng.choice([0.4, 2.0, 10.0]))
    p.starbeam_filter = int(rng.random() < 0.3)
    p.dmin_param = int(rng.choice([3, 10, 30]))
    # the reference reads array3D[1] / array3D[10] unconditionally: keep channels > 10 when it is the judge
    p.channels = int(rng.choice([16, 64, 128] if for_reference else [1, 2, 7, 11, 16, 64, 128]))
    if not for_reference:
        # the reference's `rep` is a compile-time 360 (star_shaped_search.cpp:8); the restatement and the
        # kernels take it as a parameter, and the float fast path's margin scales with it
        p.sectors = int(rng.choice([360, 360, 90, 720, 1022]))
    return p


def random_cloud(rng, n, tie_grid=0.0):
    """tie_grid > 0: x and y are snapped to multiples of it and equal planar ranges stay in."""
    kind = rng.integers(0, 3)
    if kind == 0:      # scattered returns from a ground-like sheet with steps
        x = rng.uniform(-40, 40, n)
        y = rng.uniform(-40, 40, n)
        z = -1.8 + 0.15 * (np.abs(y) > 4) + 0.02 * rng.standard_normal(n)
    elif kind == 1:    # a handful of elevation rings in random azimuth order
        rings = rng.integers(2, 40)
        elev = np.deg2rad(-25 + 23 * rng.random(rings))[rng.integers(0, rings, n)]
        az = rng.uniform(0, 2 * np.pi, n)
        h = 1.8 - 0.15 * (rng.random(n) < 0.2)
        t = h / -np.sin(elev)
        x, y, z = t * np.cos(elev) * np.cos(az), t * np.cos(elev) * np.sin(az), t * np.sin(elev)
    else:              # everything in a narrow wedge: few crowded sectors, long rings
        az = np.deg2rad(rng.uniform(20, 24, n))
        elev = np.deg2rad(rng.choice([-20.0, -12.0, -6.0], n))
        t = 1.8 / -np.sin(elev) * (1 + 0.05 * rng.random(n))
        x, y, z = t * np.cos(elev) * np.cos(az), t * np.cos(elev) * np.sin(az), t * np.sin(elev) + 0.1 * (rng.random(n) < 0.1)
    x, y, z = x.astype(np.float32), y.astype(np.float32), z.astype(np.float32)
    # a few hostile values (dropped by the ROI stage)
    bad = rng.integers(0, n, max(1, n // 200))
    x[bad[: len(bad) // 2]] = np.nan
    z[bad[len(bad) // 2:]] = 50.0
    if tie_grid:
        x = (np.round(x / np.float32(tie_grid)) * np.float32(tie_grid)).astype(np.float32)
        y = (np.round(y / np.float32(tie_grid)) * np.float32(tie_grid)).astype(np.float32)
    # x == y == 0 comes in through axis_points() only; tie-free clouds: no two equal planar ranges
    keep = ~((x == 0) & (y == 0))
    if not tie_grid:
        r = np.sqrt(x * x + y * y)
        _, first = np.unique(np.where(np.isnan(r), -1.0, r), return_index=True)
        m = np.zeros(n, bool)
        m[first] = True
        m |= np.isnan(r)
        keep &= m
    fi = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    keep &= ~((fi < 0) & (fi > -1e-5))   # stay clear of the sector-360 band
    return x[keep], y[keep], z[keep]


def axis_points(cloud, rng, count, near=0):
    """`count` points with x == y == 0 (azimuth NaN, vertical angle exactly 0 or 180 deg: the ring table's end mark) and
    `near` points almost on the axis (they can share a ring with them when `interval` is large), at random places of the
    cloud.  The axis points share one z: equal planar ranges (0) inside sector 0 are a tie the reference leaves open."""
    x, y, z = cloud
    zs = float(rng.choice([-1.8, -2.5, -1.2, 0.3]))
    ax = np.zeros(count + near, np.float32)
    ay = np.zeros(count + near, np.float32)
    az = np.full(count + near, zs, np.float32)
    if near:
        fi = rng.uniform(0, 2 * np.pi, near)
        rho = rng.uniform(0.004, 0.045, near)   # 0.13 .. 1.4 deg off the axis at 1.8 m
        ax[count:] = (rho * np.cos(fi)).astype(np.float32)
        ay[count:] = (rho * np.sin(fi)).astype(np.float32)
        az[count:] = (-1.8 + 0.2 * rng.random(near) * (rng.random(near) < 0.3)).astype(np.float32)
        r = np.sqrt(ax[count:] * ax[count:] + ay[count:] * ay[count:])
        assert len(np.unique(r)) == near   # no planar-range ties
    at = np.sort(rng.integers(0, len(x) + 1, count + near))
    order = rng.permutation(count + near)
    return tuple(np.insert(a, at, b[order]) for a, b in ((x, ax), (y, ay), (z, az)))


def case(seed, for_reference=False):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([40, 300, 3000, 20000]))
    rng3 = np.random.default_rng(seed + 9_000_011)   # (a stream of its own, as below)
    tie_grid = float(rng3.choice([1 / 64, 1 / 16, 1 / 4])) if rng3.random() < 0.4 else 0.0
    cloud, params = random_cloud(rng, n, tie_grid), random_params(rng, for_reference)
    rng2 = np.random.default_rng(seed + 7_000_003)   # (a stream of its own: the clouds of a seed stay what they were)
    if rng2.random() < 1 / 3:
        cloud = axis_points(cloud, rng2, 1 if for_reference else int(rng2.integers(1, 4)), near=int(rng2.integers(0, 6)))
    return cloud, params


def cloud_with_identical_points(seed, n, pairs):
    """A street sweep in which `pairs` points have been overwritten by copies of other points: two identical points are
    neighbours in their star sector's sorted order, the slope between them is 0 / 0 = NaN, which the walk counts and
    skips (star_shaped_search.cpp:131-132).  Returns the cloud and the indices involved (which of two identical points
    the walk marks is decided by std::sort's order of equal ranges)."""
    import oracles as O
    x, y, z = [a[:n].copy() for a in O.cfg_cloud("cfg2", seed)]
    rng = np.random.default_rng(seed)
    src = rng.choice(n, pairs, replace=False)
    dst = (src + rng.integers(1, n, pairs)) % n
    dst = np.setdiff1d(dst, src)
    src = src[:len(dst)]
    x[dst], y[dst], z[dst] = x[src], y[src], z[src]
    return (x, y, z), np.concatenate([src, dst])


def assert_equal_up_to_identical_points(lg, lb, scan, involved):
    same = lg == lb
    if not same.all():
        # a disagreement may only swap the labels of identical points
        x, y, z = scan
        bad = np.flatnonzero(~same)
        assert np.isin(bad, involved).all(), bad[:10]
        key = lambda i: (x[i].tobytes(), y[i].tobytes(), z[i].tobytes())
        for i in bad:
            twins = [j for j in involved if key(j) == key(i)]
            assert sorted(lg[twins]) == sorted(lb[twins]), (i, twins)


def killer_sector_cloud(n, dup, seed):
    """n points of ONE star sector (polar angle 10.2 .. 10.8 deg) whose planar ranges, in input order, are an input on
    which libstdc++'s std::sort reaches its depth limit (McIlroy's adversary against the real std::sort,
    oracle/stdsort_ref.cpp), a fraction `dup` of them lowered onto their neighbour in value so that equal ranges occur
    (0.05 / 0.1: the limit is still reached, tests assert it); plus a thin sweep around it so that rings and the other
    sectors exist.  Equal adversary values share their (x, y) exactly; heights differ."""
    import ctypes as C
    import os
    import oracles as O
    O.ensure_built()
    ref = C.CDLL(os.path.join(O.ORACLE_DIR, "libstdsort_ref.so"))
    ref.urf_ref_killer.argtypes = [C.c_void_p, C.c_int]
    v = np.zeros(n, np.float32)
    ref.urf_ref_killer(v.ctypes.data, n)
    rng = np.random.default_rng(n)
    j = rng.choice(n, int(dup * n), replace=False)
    v[j] = np.maximum(v[j] - 1, 0)
    rng = np.random.default_rng(seed)
    r = 4.0 + 30.0 * v / max(1.0, float(v.max()))
    fi = np.deg2rad(10.2 + 0.6 * (v * 0.6180339887 % 1.0))          # a function of the value: equal values, equal (x, y)
    xs, ys = (r * np.cos(fi)).astype(np.float32), (r * np.sin(fi)).astype(np.float32)
    rr = np.sqrt(xs * xs + ys * ys)
    order = np.argsort(v, kind="stable")                               # the float ranges must rise with the values
    assert np.all(np.diff(rr[order]) >= 0) and np.all((np.diff(rr[order]) == 0) == (np.diff(v[order]) == 0))
    zs = (-1.8 + 0.25 * rng.random(n) * (rng.random(n) < 0.5)).astype(np.float32)
    bx, by, bz = [a[::9].copy() for a in O.cfg_cloud("narrow", seed)]
    keep = ~((np.degrees(np.arctan2(by, bx)) > 9.5) & (np.degrees(np.arctan2(by, bx)) < 11.5))
    return (np.concatenate([xs, bx[keep]]), np.concatenate([ys, by[keep]]), np.concatenate([zs, bz[keep]]))
