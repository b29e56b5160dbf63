"""GPU parity tests: the HIP path, called through the C ABI (include/urf.h), against
  - the golden label vectors produced by the reference's own sources (tests/golden/),
  - oracle B on the same seeded inputs (labels, per-scan counts and every intermediate stage),
  - size-independent properties at the benchmark's full batch size.
Bar: integer labels / ring / sector ids bit-exact; floats (vertical angle, azimuth, planar range)
within 1e-5 relative -- they are in fact bit-identical because host and device share
include/urf_libm.h and every other operation is an IEEE basic operation."""
import ctypes as C
import os

import numpy as np
import pytest

import fuzz
import oracles as O
import urban_road_filter_amd as u
from golden.make_golden import CASES, case_params, cloud_sha
from hipmem import DevBuf

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FTOL = 1e-5


def close(a, b):
    return np.abs(a.astype(np.float64) - b) <= FTOL * np.maximum(1.0, np.abs(b))


@pytest.fixture(scope="module")
def ctx_big():
    c = u.Context(128 * 4096, 1)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_hooks():
    """A context in liburf_hip_test.so (include/urf_test_hooks.h: debug flags, device self tests)."""
    c = u.Context(64 * 2048, 1, hooks=True)
    yield c
    c.close()


def info_equal(ig, ib):
    return all(getattr(ig, k) == ib[k] for k in ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10"))


@pytest.mark.parametrize("name,cfg,seed,tweak", CASES, ids=[c[0] for c in CASES])
def test_golden_cases(ctx_big, name, cfg, seed, tweak):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    p = case_params(cfg, tweak)
    x, y, z = O.case_cloud(cfg, seed, g)
    assert cloud_sha(x, y, z) == str(g["cloud_sha"])
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(lg & O.MASK_NO_RING, g["labels"]), "differs from the reference's labels"
    lb, ib, _ = O.run_b(x, y, z, p)
    assert np.array_equal(lg, lb)
    assert info_equal(ig, ib)


@pytest.mark.parametrize("cfg,seed,tweak", [("cfg2", 2, {}), ("narrow", 3, {"xDirection": 2}),
                                            ("cfg5", 2, {}), ("default_roi", 3, {"starbeam_filter": 1}),
                                            ("cfg1", 4, {})])
def test_every_stage(ctx_big, cfg, seed, tweak):
    p = case_params(cfg, tweak)
    x, y, z = O.cfg_cloud(cfg, seed)
    n = len(x)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    ctx_big.enable_stage_capture(1)
    try:
        lg, ig = ctx_big.classify_xyz(x, y, z)
        roi = (lb & u.FLAG_ROI) != 0
        ring = (lb & u.FLAG_RING) != 0
        va = ctx_big.read_stage(u.STAGE_VALPHA, n)
        assert close(va[roi], st["valpha"][roi]).all() and (va[~roi] < 0).all()
        assert np.array_equal(va[roi], st["valpha"][roi])                      # in fact bit-identical
        assert np.array_equal(ctx_big.read_stage(u.STAGE_ANGLE_TABLE, n), st["angle_table"])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_RING, n), st["ring"])
        if p.star_shaped_method:
            assert np.array_equal(ctx_big.read_stage(u.STAGE_SECTOR, n), st["sector"])
        az = ctx_big.read_stage(u.STAGE_AZIMUTH, n)
        d2 = ctx_big.read_stage(u.STAGE_RANGE2D, n)
        assert close(az[ring], st["azimuth"][ring]).all() and close(d2[ring], st["range2d"][ring]).all()
        assert np.array_equal(az[ring], st["azimuth"][ring]) and np.array_equal(d2[ring], st["range2d"][ring])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, n), st["detect"])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_MAXDIST, n), st["max_dist"])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_QUADRANTS, n), st["quadrants"])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_BEAM_STOP, n), st["beam_stop"])
        assert np.array_equal(lg, lb) and info_equal(ig, ib)
    finally:
        ctx_big.enable_stage_capture(0)
    # the production configuration: ring, sector and window membership are decided on float
    # approximations with margins; every integer stage must still be the reference's (capture
    # mode 2 = the production decisions, with ring and sector of every input point recorded)
    ctx_big.enable_stage_capture(2)
    try:
        lg, ig = ctx_big.classify_xyz(x, y, z)
        assert np.array_equal(ctx_big.read_stage(u.STAGE_RING, n), st["ring"])
        if p.star_shaped_method:
            assert np.array_equal(ctx_big.read_stage(u.STAGE_SECTOR, n), st["sector"])
        assert np.array_equal(lg, lb) and info_equal(ig, ib)
    finally:
        ctx_big.enable_stage_capture(0)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, n), st["detect"])
    assert np.array_equal(ctx_big.read_stage(u.STAGE_MAXDIST, n), st["max_dist"])
    assert np.array_equal(ctx_big.read_stage(u.STAGE_QUADRANTS, n), st["quadrants"])
    assert np.array_equal(ctx_big.read_stage(u.STAGE_BEAM_STOP, n), st["beam_stop"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def run_batch(ctx, scans, p, ragged=False):
    """Device-resident batch through urf_classify_batch_soa(_ragged); returns labels per scan + infos."""
    lens = [len(s[0]) for s in scans]
    X = np.concatenate([s[0] for s in scans]) if scans else np.zeros(0, np.float32)
    Y = np.concatenate([s[1] for s in scans]) if scans else np.zeros(0, np.float32)
    Z = np.concatenate([s[2] for s in scans]) if scans else np.zeros(0, np.float32)
    dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
    dl = DevBuf(len(X))
    dl.fill(0xEE)
    di = DevBuf(32 * len(scans))
    ctx.set_params(p)
    if ragged:
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        do = DevBuf.from_numpy(offs)
        ctx.classify_batch_soa_ragged(dx, dy, dz, do, max(lens), len(scans), dl, di)
    else:
        assert len(set(lens)) == 1
        ctx.classify_batch_soa(dx, dy, dz, lens[0], len(scans), dl, di)
    ctx.synchronize()
    L = dl.to_numpy(np.uint8)
    infos = di.to_numpy(np.uint32).reshape(len(scans), 8)
    out, pos = [], 0
    for n in lens:
        out.append(L[pos:pos + n])
        pos += n
    return out, infos


def test_ragged_batch_inside_a_larger_buffer(ctx_big):
    """The scans of a ragged batch may lie anywhere in the caller's arrays: offsets that do not start
    at 0, gaps between scans, and a scan longer than max_len (cut there).  Scratch memory is indexed
    by scan, not by these offsets."""
    p = O.cfg_params("cfg2")
    scans = [O.cfg_cloud("cfg2", 30 + k) for k in range(3)]
    scans = [tuple(a[:n].copy() for a in s) for s, n in zip(scans, (40000, 131072, 9000))]
    lead, gap = 777, 1234
    X = np.full(lead + sum(len(s[0]) for s in scans) + 2 * gap + 50, 7.5, np.float32)
    Y, Z = X.copy(), X.copy()
    offs, pos = [], lead
    for x, y, z in scans:
        offs.append(pos)
        X[pos:pos + len(x)], Y[pos:pos + len(x)], Z[pos:pos + len(x)] = x, y, z
        pos += len(x) + gap
    # offsets[s+1] - offsets[s] is the length: the gap belongs to the previous scan as junk points
    # (7.5, 7.5, 7.5: outside the z ROI), except for the last scan
    offs.append(pos - gap)
    dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
    do = DevBuf.from_numpy(np.array(offs, np.uint32))
    dl = DevBuf(len(X))
    dl.fill(0xEE)
    with u.Context(131072 + gap, 3, params=p) as ctx:
        ctx.classify_batch_soa_ragged(dx, dy, dz, do, 131072 + gap, 3, dl, None)
        ctx.synchronize()
        L = dl.to_numpy(np.uint8)
        assert (L[:lead] == 0xEE).all() and (L[offs[3]:] == 0xEE).all()        # nothing outside the scans is touched
        for k, (x, y, z) in enumerate(scans):
            n = offs[k + 1] - offs[k]
            xx, yy, zz = X[offs[k]:offs[k] + n], Y[offs[k]:offs[k] + n], Z[offs[k]:offs[k] + n]
            lb, _, _ = O.run_b(xx, yy, zz, p)
            assert np.array_equal(L[offs[k]:offs[k] + n], lb), k
        # a scan longer than max_len is cut at max_len: labels behind it stay untouched
        dl.fill(0xEE)
        ctx.classify_batch_soa_ragged(dx, dy, dz, do, 20000, 3, dl, None)
        ctx.synchronize()
        L = dl.to_numpy(np.uint8)
        for k in range(3):
            n = min(offs[k + 1] - offs[k], 20000)
            lb, _, _ = O.run_b(X[offs[k]:offs[k] + n], Y[offs[k]:offs[k] + n], Z[offs[k]:offs[k] + n], p)
            assert np.array_equal(L[offs[k]:offs[k] + n], lb), k
            assert (L[offs[k] + n:offs[k + 1]] == 0xEE).all()


def check_against_b(labels, infos, scans, p):
    for k, (x, y, z) in enumerate(scans):
        lb, ib, _ = O.run_b(x, y, z, p)
        assert np.array_equal(labels[k], lb), "scan %d" % k
        keys = ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10")
        got = dict(zip(keys, infos[k][:7]))
        assert got == {f: ib[f] for f in keys}, "scan %d" % k
        assert infos[k][7] == 0, "scan %d: NaN azimuths counted on a sweep without x == y == 0 points" % k


def test_uniform_batch_of_distinct_scans():
    p = O.cfg_params("cfg2")
    scans = [O.cfg_cloud("cfg2" if s % 2 else "narrow", 10 + s) for s in range(6)]
    with u.Context(64 * 2048, 6) as ctx:
        labels, infos = run_batch(ctx, scans, p)
        check_against_b(labels, infos, scans, p)


def test_batch_of_eleven_scans():
    """k_label maps its workgroups XCD-aware in groups of eight scans; the scans of an incomplete group
    keep the plain mapping.  Eleven scans: one complete group and three scans outside it, one of them
    failing the 30-point threshold."""
    p = O.cfg_params("cfg2")
    scans = [O.cfg_cloud("cfg2" if s % 3 else "narrow", 40 + s) for s in range(11)]
    few = tuple(a.copy() for a in scans[9])
    few[0][29:] = 1.0e6   # all but 29 points far outside the region of interest
    scans[9] = few
    with u.Context(64 * 2048, 11) as ctx:
        labels, infos = run_batch(ctx, scans, p)
        check_against_b(labels, infos, scans, p)
        assert infos[9][0] == 1 and not labels[9].any()


def test_ragged_batch_with_empty_tiny_and_partial_scans():
    p = O.cfg_params("cfg2")
    full = O.cfg_cloud("cfg2", 21)
    part = tuple(a[:50000].copy() for a in O.cfg_cloud("cfg2", 22))
    odd = tuple(a[:4097].copy() for a in O.cfg_cloud("narrow", 23))     # one point into the second tile
    tiny = tuple(a[:29].copy() for a in full)                          # < 30 ROI points
    empty = tuple(np.zeros(0, np.float32) for _ in range(3))
    flat16 = O.cfg_cloud("cfg1", 24)                                    # 16 rings with channels = 64
    scans = [part, empty, full, tiny, odd, flat16]
    with u.Context(64 * 2048, len(scans)) as ctx:
        labels, infos = run_batch(ctx, scans, p, ragged=True)
        check_against_b(labels, infos, scans, p)
        assert infos[1][0] == 1 and infos[3][0] == 1 and not labels[3].any()   # URF_TOO_FEW_POINTS, nothing published


def test_pointcloud2_layouts(ctx_big):
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("cfg2", 31)
    n = len(x)
    lb, ib, _ = O.run_b(x, y, z, p)
    ctx_big.set_params(p)
    # Ouster-like record: x y z pad intensity t reflectivity ring ... point_step 48
    rec = np.zeros((n, 48), np.uint8)
    rec[:, 0:4] = x.view(np.uint8).reshape(n, 4)
    rec[:, 4:8] = y.view(np.uint8).reshape(n, 4)
    rec[:, 8:12] = z.view(np.uint8).reshape(n, 4)
    rec[:, 16:48] = 0xAB
    lg, ig = ctx_big.classify_pc2(rec, n, 48, 0, 4, 8)
    assert np.array_equal(lg, lb) and info_equal(ig, ib)
    # permuted, unaligned fields: point_step 23, z at 1, x at 9, y at 17
    rec = np.full((n, 23), 0x5A, np.uint8)
    rec[:, 1:5] = z.view(np.uint8).reshape(n, 4)
    rec[:, 9:13] = x.view(np.uint8).reshape(n, 4)
    rec[:, 17:21] = y.view(np.uint8).reshape(n, 4)
    lg, ig = ctx_big.classify_pc2(rec, n, 23, 9, 17, 1)
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_roi_edge_cases(ctx_big):
    p = O.cfg_params("cfg2")
    x, y, z = [a.copy() for a in O.cfg_cloud("cfg2", 41)]
    x[100] = np.nan
    y[101] = np.inf
    x[200] = y[200] = z[200] = 0.0                 # "no return"
    x[300], y[300], z[300] = 1.0, 1.0, -2.0        # x + y + z == 0 is dropped (lidar_segmentation.cpp:111)
    z[400:500] = 0.5                               # above the z ROI
    lb, ib, _ = O.run_b(x, y, z, p)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(lg, lb) and info_equal(ig, ib)
    assert not lg[[100, 101, 200, 300]].any() and not lg[400:500].any()
    # everything outside the ROI
    lg, ig = ctx_big.classify_xyz(x, y, np.full_like(z, 10.0))
    assert ig.status == 1 and ig.n_roi == 0 and not lg.any()


def test_thirty_point_threshold(ctx_big):
    p = O.cfg_params("cfg2")
    x, y, z = [a[:4096].copy() for a in O.cfg_cloud("cfg2", 1)]
    z[29:] = 5.0
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert ig.status == 1 and ig.n_roi == 29 and not lg.any()
    z[29] = -1.8
    lb, ib, _ = O.run_b(x, y, z, p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert ig.status == 0 and ig.n_roi == 30 and np.array_equal(lg, lb) and info_equal(ig, ib)


@pytest.mark.parametrize("tweak", [{"channels": 32}, {"channels": 7, "interval": 0.5}, {"interval": 0.4},
                                   {"curbPoints": 1}, {"curbPoints": 30}, {"beamZone": 100.0}, {"beamZone": 10.25},
                                   {"x_zero_method": 0, "star_shaped_method": 0}, {"z_zero_method": 0, "x_zero_method": 0},
                                   {"dmin_param": 3, "kdist_param": 0.4, "angleFilter3": 5.0},
                                   {"angleFilter1": 179.0, "angleFilter2": 10.0, "curbHeight": 0.01}])
def test_parameter_corners(ctx_big, tweak):
    """Ring table overflow (more rings than channels, lidar_segmentation.cpp:191), merged rings,
    extreme curb_points / beamZone, detectors switched off one by one."""
    p = case_params("narrow", tweak)
    x, y, z = O.cfg_cloud("narrow", 51)
    lb, ib, _ = O.run_b(x, y, z, p)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def crowded_cloud(n_pts, rings=3, seed=5):
    """All points in a handful of sectors and rings: sector sizes far beyond one LDS tile
    (exercises the 512 < n <= 2048 and the global-memory sort paths) and rings longer than a tile."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(-20.0 + 4.0 * (np.arange(n_pts) % rings))
    az = np.deg2rad(10.3 + 0.4 * rng.random(n_pts))          # inside sectors 10 / 11
    t = 1.8 / -np.sin(elev) * (1 + 0.02 * rng.random(n_pts))
    x = (t * np.cos(elev) * np.cos(az)).astype(np.float32)
    y = (t * np.cos(elev) * np.sin(az)).astype(np.float32)
    z = (t * np.sin(elev)).astype(np.float32)
    # make the planar ranges unique (ties are unspecified in the reference)
    r = np.sqrt(x * x + y * y)
    _, first = np.unique(r, return_index=True)
    keep = np.sort(first)
    return x[keep], y[keep], z[keep]


@pytest.mark.parametrize("n_pts", [1500, 6000, 40000])
def test_crowded_sectors(ctx_big, n_pts):
    p = O.cfg_params("cfg2")
    p.interval = 1.0
    x, y, z = crowded_cloud(n_pts)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_many_star_hits_on_one_ring(ctx_big):
    """With a wide `interval` the sweep's 64 beams merge into a handful of rings, so that one ring collects
    more star-shaped hits than k_ring keeps in LDS (URF_RING_HITS = 62) and every chunk rescans the scan's
    hits instead."""
    p = O.cfg_params("cfg2")
    p.interval = 4.0
    x, y, z = O.cfg_cloud("cfg2", 31)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    assert ib["n_rings"] <= 6
    ring_of_hit = st["ring"][(st["detect"] & 1) != 0]
    assert np.bincount(ring_of_hit[ring_of_hit >= 0]).max() > 62
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_mid_size_sector_scattered_over_many_tiles(ctx_big):
    """A sector of 385..2048 points whose points are spread over ten input tiles: k_star_sort_mid then
    cannot use the two-run description of k_index and builds the sector's run list from the per-tile
    tables; the small sectors around it take the run-list path of k_star_sort_small."""
    p = O.cfg_params("cfg2")
    p.interval = 1.0
    xc, yc, zc = crowded_cloud(1400, seed=11)           # ~1400 points in sectors 10 / 11
    xs, ys, zs = O.cfg_cloud("narrow", 77)
    keep = np.arange(0, len(xs), 7)                      # a thinned sweep: small sectors everywhere
    xs, ys, zs = xs[keep], ys[keep], zs[keep]
    x, y, z = np.concatenate([xc, xs]), np.concatenate([yc, ys]), np.concatenate([zc, zs])
    perm = np.random.default_rng(12).permutation(len(x))   # unorganised: every sector meets every tile
    x, y, z = x[perm], y[perm], z[perm]
    assert len(x) > 8 * 2048
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


@pytest.mark.parametrize("twins", [False, True], ids=["other_height", "twins"])
@pytest.mark.parametrize("n_pts", [300, 1500, 6000])
def test_equal_ranges_follow_std_sort(ctx_big, n_pts, twins):
    """Exact planar-range ties inside a sector are ordered as libstdc++'s std::sort orders them
    (star_shaped_search.cpp:109; oracle/urf_stdsort.h pins the algorithm against the real one, k_star_ties follows it
    on the device).  Duplicated (x, y) with another height -- the slope between the two is +-inf, the sign depends on
    the order -- on all three sizes: one wave's LDS (<= 512 points per sector), the big instance's LDS (<= 2048) and
    global memory.  twins: the duplicates keep their height too -- the walk's arithmetic is then the same in any order and only
    WHICH of the twins stands where the walk stops depends on it: k_star_ties' second pass, behind the walk."""
    p = O.cfg_params("cfg2")
    p.interval = 1.0
    x, y, z = crowded_cloud(n_pts, seed=9)
    rng = np.random.default_rng(4)
    dup = rng.integers(0, len(x), len(x) // 3)
    x = np.concatenate([x, x[dup]])
    y = np.concatenate([y, y[dup]])
    z = np.concatenate([z, z[dup] if twins else (z[dup] + 0.2 * rng.random(len(dup))).astype(np.float32)])   # same range; other height or the same
    perm = rng.permutation(len(x))
    x, y, z = x[perm], y[perm], z[perm]
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)
    # the same through the batch entry point (the full kernel sequence at once), twice in one batch
    with u.Context(len(x), 2, params=p) as ctx:
        labels, infos = run_batch(ctx, [(x, y, z), (x[::-1].copy(), y[::-1].copy(), z[::-1].copy())], p)
        assert np.array_equal(labels[0], lb)
        lr, _, _ = O.run_b(x[::-1].copy(), y[::-1].copy(), z[::-1].copy(), p)
        assert np.array_equal(labels[1], lr)


@pytest.mark.parametrize("n,dup", [(364, 0.0), (364, 0.1), (500, 0.05), (1024, 0.1), (2048, 0.05), (4000, 0.05)])
def test_std_sort_depth_limit_on_the_device(ctx_big, n, dup):
    """A sector whose ranges are an adversarial input for libstdc++'s introsort: the depth limit 2 * floor(log2 n) is
    reached and the segment is heap sorted (stl_algo.h __partial_sort), which k_star_ties does statement by statement
    on one lane; with equal ranges in it, whose final order is the heap's."""
    import ctypes as C
    x, y, z = fuzz.killer_sector_cloud(n, dup, seed=n)
    if dup == 0.0:   # tie-free: nothing flags the sector -- one duplicated pair in front of the walk's stop does
        x, y, z = np.concatenate([x[:1], x]), np.concatenate([y[:1], y]), np.concatenate([z[:1] + np.float32(0.3), z])
    p = O.cfg_params("cfg2")
    p.interval = 1.0
    hs = O.oracle_b().urf_oracle_std_sort_heap_sorts
    hs.restype = C.c_long
    before = hs()
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    if dup in (0.05, 0.1) and n != 500 and n != 2048:
        assert hs() > before
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_sensor_like_batch_and_callback_sequence():
    """Sweeps as a driver delivers them (ties in every star sector).  Batch: 48 sweeps = 17 280 sectors, k_star_ties'
    64-sectors-per-wave scan of the flags (a smaller batch takes one workgroup per sector).  Callback path: the short launch
    sequence has no k_star_ties -- the first sweep comes back void and is run again with it, every later one takes it at
    once (urf_callback_path_state: bit 4)."""
    p = O.cfg_params("sensor")
    clouds = [O.cfg_cloud("sensor" if k % 3 else "sensor_narrow", 20 + k) for k in range(6)]
    ref = [O.run_b(*c, p) for c in clouds]
    n = len(clouds[0][0])
    with u.Context(n, 48, params=p) as ctx:
        labels, infos = run_batch(ctx, [clouds[k % 6] for k in range(48)], p)
        for k in range(48):
            assert np.array_equal(labels[k], ref[k % 6][0]), k
        labels, infos = run_batch(ctx, clouds[:5], p)
        for k in range(5):
            assert np.array_equal(labels[k], ref[k][0]), k
        # tie-free sweeps in the same context afterwards: nothing flagged, nothing left over from the flags of the call before
        free = [O.cfg_cloud("cfg2", 30 + k) for k in range(5)]
        labels, infos = run_batch(ctx, free, p)
        for k in range(5):
            assert np.array_equal(labels[k], O.run_b(*free[k], p)[0]), k
    with u.Context(n, 4, params=p) as ctx:
        free = O.cfg_cloud("cfg2", 31)
        lg, ig = ctx.classify_xyz(*free)
        assert np.array_equal(lg, O.run_b(*free, p)[0]) and ctx.callback_path_state() == (0, 1 | 8)
        for k in range(6):
            lg, ig = ctx.classify_xyz(*clouds[k])
            assert np.array_equal(lg, ref[k][0]) and info_equal(ig, ref[k][1])
        assert ctx.callback_path_state() == (1, 1 | 8 | 16)   # (the first sweep that needed either pass of k_star_ties)
        lg, ig = ctx.classify_xyz(*free)
        assert np.array_equal(lg, O.run_b(*free, p)[0])
    # urf_callback_path_preset: a node that knows its sensor puts k_star_ties into the sequence ahead of the first sweep -- none is run twice
    with u.Context(n, 4, params=p) as ctx:
        ctx.callback_path_preset(16)
        assert ctx.callback_path_state() == (0, 1 | 8 | 16)
        for k in range(6):
            lg, ig = ctx.classify_xyz(*clouds[k])
            assert np.array_equal(lg, ref[k][0]) and info_equal(ig, ref[k][1])
        assert ctx.callback_path_state() == (0, 1 | 8 | 16)
        ctx.callback_path_preset(2 | 4)
        assert ctx.callback_path_state() == (0, 1 | 2 | 4 | 8 | 16)
        lg, ig = ctx.classify_xyz(*clouds[0])
        assert np.array_equal(lg, ref[0][0])
        with pytest.raises(Exception):
            ctx.callback_path_preset(1)   # (the speculation bits are not presets)


def test_star_sort_paths(ctx_hooks):
    ctx_big = ctx_hooks
    """k_star_sort_small has a distribution-sort fast path and a general path (in-register block
    sorts merged by ranking).  (a) force the general path on a normal sweep; (b) a cloud whose
    ranges are so clustered that buckets overflow and the kernel falls back by itself."""
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("narrow", 91)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    ctx_big.set_debug_flags(4)
    try:
        lg, ig = ctx_big.classify_xyz(x, y, z)
    finally:
        ctx_big.set_debug_flags(0)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)
    # (b) 3 rings; in sector 10: 300 points packed within a few hundred ulps of r = 10 m plus two far
    # outliers, so that one of the 512 range buckets receives far more than 64 keys
    rng = np.random.default_rng(3)
    n = 400
    az = np.deg2rad(10.2 + 0.6 * rng.random(n))
    r = (10.0 + 1e-5 * rng.random(n)).astype(np.float32)
    r[:2] = [3.0, 60.0]
    elev = np.deg2rad(-20.0 + 4.0 * (np.arange(n) % 3))
    xs = (r * np.cos(az)).astype(np.float32)
    ys = (r * np.sin(az)).astype(np.float32)
    zs = (-1.8 + 0.3 * rng.random(n) * (np.arange(n) % 7 == 0)).astype(np.float32)
    rr = np.sqrt(xs * xs + ys * ys)
    _, first = np.unique(rr, return_index=True)
    keep = np.sort(first)
    xs, ys, zs = xs[keep], ys[keep], zs[keep]
    p.interval = 5.0
    lb, ib, st = O.run_b(xs, ys, zs, p, debug=True)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(xs, ys, zs)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(xs)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_ring_table_zero_sentinel(ctx_big):
    """A point straight below the sensor has vertical angle exactly 0, which the reference's ring
    table treats as its end-of-table mark (lidar_segmentation.cpp:176), and an azimuth of NaN, which its
    per-ring quicksort parks at an input-order-dependent place where the beam scans then stop
    (:70-93, blind_spots.cpp:107,146,216,255).  Ring assignment, labels, counters and the published
    order follow the reference through all of it (k_nan_rings)."""
    p = O.cfg_params("cfg2")
    x, y, z = [a[:8192].copy() for a in O.cfg_cloud("cfg2", 61)]
    x[3] = y[3] = 0.0
    z[3] = -1.8
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    ctx_big.enable_stage_capture(2)
    try:
        lg, ig = ctx_big.classify_xyz(x, y, z)
        assert np.array_equal(ctx_big.read_stage(u.STAGE_ANGLE_TABLE, len(x)), st["angle_table"])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_RING, len(x)), st["ring"])
    finally:
        ctx_big.enable_stage_capture(0)
    assert np.array_equal(lg, lb) and info_equal(ig, ib)
    assert ig.n_nan_azimuth == int(st["ring"][3] >= 0)   # such points are counted (include/urf.h)
    x2, y2, z2 = [a.copy() for a in (x, y, z)]
    x2[100:103] = 0.0
    y2[100:103] = 0.0
    z2[100:103] = z2[100]   # (identical points: equal planar ranges inside a sector, ordered as std::sort leaves them)
    lb2, ib2, st2 = O.run_b(x2, y2, z2, p, debug=True)
    lg2, ig2 = ctx_big.classify_xyz(x2, y2, z2)
    assert ig2.n_nan_azimuth == int((st2["ring"][[3, 100, 101, 102]] >= 0).sum())
    assert info_equal(ig2, ib2)
    assert np.array_equal(lg2, lb2)


@pytest.mark.parametrize("pairs", [40, 1500])
def test_nan_slopes_single_sweep(ctx_big, pairs):
    """NaN slopes in one sweep: the three-wave walk of the callback path (k_star_walk_few) hands the sectors over to
    the sequential walk at the chunk in which a NaN mean shows up."""
    p = O.cfg_params("cfg2")
    for seed in (201, 202, 203):
        scan, involved = fuzz.cloud_with_identical_points(seed, 64 * 2048, pairs)
        lb, ib, _ = O.run_b(*scan, p)
        ctx_big.set_params(p)
        lg, ig = ctx_big.classify_xyz(*scan)
        assert info_equal(ig, ib), seed
        assert np.array_equal(lg, lb)   # (r5: which of two identical points carries the mark is std::sort's order, followed exactly)


def test_nan_slopes_in_a_batch():
    """The same through the batch walk (k_star_walk: more than URF_WALK_FEW_SCANS = 32 scans) and through the
    three-wave walk with several scans in the grid (8 scans)."""
    p = O.cfg_params("cfg2")
    n = 16 * 2048
    made = [fuzz.cloud_with_identical_points(300 + k, n, 25 if k % 2 else 400) for k in range(40)]
    scans = [m[0] for m in made]
    want = [O.run_b(*sc, p) for sc in scans]
    with u.Context(n, 40, params=p) as ctx:
        for count in (40, 8):
            labels, infos = run_batch(ctx, scans[:count], p)
            for k in range(count):
                lb, ib, _ = want[k]
                assert tuple(infos[k][:7]) == tuple(ib[f] for f in ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10")), k
                assert np.array_equal(labels[k], lb), k


def same_order(got, want, st=None):
    """r5: the published order is the reference's, the order of bit-identical azimuths included (its Lomuto quicksort is run
    literally for a ring that holds such points: k_ring_order)."""
    return np.array_equal(got, want)


def nan_ring_cloud(seed, n_axis, n_near):
    """A short organised sweep plus `n_axis` points exactly on the sensor's axis and `n_near` points almost on it,
    strewn over the input.  With a wide `interval` they share ONE ring -- sorted ring 0 -- whose azimuth-sorted array
    then holds NaN entries between real ones."""
    from fuzz import axis_points
    rng = np.random.default_rng(seed)
    base = tuple(a[:16384].copy() for a in O.cfg_cloud("narrow" if seed % 2 else "cfg2", 400 + seed))
    return axis_points(base, rng, n_axis, near=n_near)


@pytest.mark.parametrize("seed", range(16))
def test_rings_with_nan_azimuths_follow_the_reference(ctx_big, seed):
    """Deviation D5 of earlier rounds, closed: a ring that holds points with x == y == 0 is sorted by the reference's own
    Lomuto quicksort (the NaN azimuths land where THAT leaves them) and its beam scans end at the first NaN they meet.
    Labels, counters, beam stops, published order and marker points equal oracle B, which equals the reference's
    binary on such input (tests/test_fuzz_cpu.py)."""
    p = O.cfg_params("cfg2")
    p.interval = [1.5, 1.5, 0.5, 0.18][seed % 4]
    p.curbPoints = [5, 2, 5, 9][(seed // 4) % 4]
    p.curbHeight = 0.02
    p.channels = 64 if seed % 3 == 0 else 72   # (64: the sweep's own rings fill the table unless an axis point comes first)
    x, y, z = nan_ring_cloud(seed, 1 + seed % 4, [0, 7, 40, 150][(seed // 2) % 4])
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    axis = (x == 0) & (y == 0)
    # (several axis points are identical points of star sector 0: which of them the walk marks is std::sort's order of equal ranges)
    assert np.array_equal(lg, lb), "%d labels differ" % int((lg != lb).sum())
    assert ig.n_rings == ib["n_rings"] and ig.n_road == ib["n_road"] and ig.n_ring_pts == ib["n_ring_pts"]
    assert ig.n_nan_azimuth == int((st["ring"][axis] >= 0).sum())
    assert np.array_equal(ctx_big.read_stage(u.STAGE_BEAM_STOP, len(x)), st["beam_stop"])
    if True:
        road, curb, prob = ctx_big.ordered_indices(len(x))
        # (a wide `interval` merges lasers into one ring: points of one firing then share an azimuth to the bit)
        assert same_order(road, st["road_order"], st) and same_order(curb, st["curb_order"], st)
        assert same_order(prob, st["ring10_order"], st)
        mg, mw = ctx_big.marker_points(), st["marker_pts"]
        assert mg.shape == mw.shape and np.array_equal(mg, mw)


def late_ring_cloud(n=60000, late_at=40000, seed=5):
    """8 rings from the first firing on, a 9th one that shows up only `late_at` points into the sweep."""
    rng = np.random.default_rng(seed)
    i = np.arange(n)
    elev = np.deg2rad(-22.0 + 2.0 * (i % 8))
    late = (i >= late_at) & (i % 97 == 0)
    elev[late] = np.deg2rad(-4.0)
    az = i * (2 * np.pi / n) + 1e-3
    t = 1.8 / -np.sin(elev) * (1 + 1e-4 * rng.random(n))
    x, y, z = t * np.cos(elev) * np.cos(az), t * np.cos(elev) * np.sin(az), t * np.sin(elev)
    x, y, z = x.astype(np.float32), y.astype(np.float32), z.astype(np.float32)
    r = np.sqrt(x * x + y * y)
    _, first = np.unique(r, return_index=True)   # no planar-range ties inside a sector
    keep = np.zeros(n, bool)
    keep[first] = True
    return x[keep], y[keep], z[keep]


def test_speculative_ring_table_is_repaired():
    """k_ring_table stops looking for new rings after 8192 quiet points and lets k_split check the
    rest; a ring that shows up later than that makes k_split raise the scan's redo flag, the table
    is rebuilt the long way and the scan split again -- in the same call.  Afterwards the context no
    longer speculates; results stay the reference's either way."""
    p = O.cfg_params("cfg2")
    x, y, z = late_ring_cloud()
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    assert ib["n_rings"] == 9
    ok = O.cfg_cloud("cfg2", 88)
    ok = tuple(a[:len(x)].copy() for a in ok)
    lb_ok, ib_ok, _ = O.run_b(*ok, p)
    with u.Context(len(x), 3, params=p) as ctx:
        labels, infos = run_batch(ctx, [ok, (x, y, z), ok], p)          # only the middle scan needs the repair
        assert np.array_equal(labels[1], lb) and infos[1][2] == 9
        assert np.array_equal(labels[0], lb_ok) and np.array_equal(labels[2], lb_ok)
        labels, infos = run_batch(ctx, [(x, y, z), ok, (x, y, z)], p)   # (no speculation any more)
        assert np.array_equal(labels[0], lb) and np.array_equal(labels[2], lb) and np.array_equal(labels[1], lb_ok)
    with u.Context(len(x), 1, params=p) as ctx:                         # the callback path: captured graph, then re-captured
        for rep in range(3):
            lg, ig = ctx.classify_xyz(x, y, z)
            assert np.array_equal(lg, lb) and info_equal(ig, ib)
            lg, ig = ctx.classify_xyz(*ok)
            assert np.array_equal(lg, lb_ok)


def test_a_long_ring_with_nan_azimuths(ctx_big):
    """More points on the NaN ring than k_nan_rings keeps in LDS (6 144): the literal quicksort runs in global memory."""
    rng = np.random.default_rng(77)
    n = 9000
    fi = rng.uniform(0, 2 * np.pi, n)
    rho = np.linspace(0.004, 0.044, n)[rng.permutation(n)]   # distinct planar ranges
    x, y = (rho * np.cos(fi)).astype(np.float32), (rho * np.sin(fi)).astype(np.float32)
    z = (-1.8 + 0.2 * rng.random(n) * (rng.random(n) < 0.3)).astype(np.float32)
    r = np.sqrt(x * x + y * y)
    _, first = np.unique(r, return_index=True)
    keep = np.sort(first)
    x, y, z = x[keep], y[keep], z[keep]
    base = tuple(a[:4096].copy() for a in O.cfg_cloud("cfg2", 431))
    x, y, z = (np.concatenate([b, v]) for b, v in zip(base, (x, y, z)))
    x, y, z = (np.insert(a, 6000, v) for a, v in ((x, 0.0), (y, 0.0), (z, -1.8)))
    p = O.cfg_params("cfg2")
    p.interval, p.curbHeight, p.channels = 1.5, 0.02, 72
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    nan_ring = st["ring"][6000]
    assert nan_ring >= 0 and (st["ring"] == nan_ring).sum() > 6144
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(lg, lb) and info_equal(ig, ib) and ig.n_nan_azimuth == 1
    road, curb, prob = ctx_big.ordered_indices(len(x))
    assert same_order(road, st["road_order"], st) and same_order(curb, st["curb_order"], st)
    for got, want in ((road, st["road_order"]), (curb, st["curb_order"])):
        assert np.array_equal(got[st["ring"][got] == nan_ring], want[st["ring"][want] == nan_ring])


def test_ring_count_hint_and_its_failure():
    """Second speculation of k_ring_table (r4): the walk also stops once the table holds as many rings as the previous call
    found -- a stream of sweeps from one sensor shows the same rings sweep after sweep.  (i) default-ROI sweeps one after the
    other: the hint is in use from the second one on, labels and ring tables stay the reference's; (ii) a sweep whose 9th
    ring shows up 3 000 points in, behind a sweep with 8 rings: the hint stops the walk too early, k_split notices, the scan
    is repaired (batch call) / run again (callback path), and only the HINT is switched off: the look-ahead speculation
    stays."""
    q = O.cfg_params("default_roi")
    clouds = [O.cfg_cloud("default_roi", 90 + k) for k in range(4)]
    with u.Context(64 * 2048, 1, params=q) as ctx:
        for rep in range(2):
            for c in clouds:
                lb, ib, _ = O.run_b(*c, q)
                lg, ig = ctx.classify_xyz(*c)
                assert np.array_equal(lg, lb) and info_equal(ig, ib)
        assert ctx.callback_path_state() == (0, 1 | 8)   # nothing was run again, both speculations still on
    p = O.cfg_params("cfg2")
    late = late_ring_cloud(late_at=3000)
    lb_late, ib_late, _ = O.run_b(*late, p)
    assert ib_late["n_rings"] == 9
    x, y, z = late
    el = np.degrees(np.arctan2(-z, np.hypot(x, y)))
    eight = tuple(a[el > 5.0].copy() for a in late)   # the same sweep without the points of the late ring (elevation -4 deg)
    lb8, ib8, _ = O.run_b(*eight, p)
    assert ib8["n_rings"] == 8
    with u.Context(len(x), 2, params=p) as ctx:
        labels, infos = run_batch(ctx, [eight, eight], p)
        assert np.array_equal(labels[0], lb8) and infos[1][2] == 8
        labels, infos = run_batch(ctx, [late, eight], p, ragged=True)      # the hint (8) ends scan 0's walk in front of its 9th ring: repaired
        assert np.array_equal(labels[0], lb_late) and infos[0][2] == 9 and np.array_equal(labels[1], lb8)
        labels, infos = run_batch(ctx, [eight, late], p, ragged=True)      # (the hint is off now, the look-ahead finds the ring 3 000 points in)
        assert np.array_equal(labels[1], lb_late) and np.array_equal(labels[0], lb8)
        assert ctx.callback_path_state()[1] & 9 == 1
    with u.Context(len(x), 1, params=p) as ctx:               # the callback path: voided, run again without the hint
        lg, ig = ctx.classify_xyz(*eight)
        assert np.array_equal(lg, lb8) and ctx.callback_path_state() == (0, 1 | 8)
        lg, ig = ctx.classify_xyz(*late)
        assert np.array_equal(lg, lb_late) and info_equal(ig, ib_late)
        assert ctx.callback_path_state() == (1, 1)
        lg, ig = ctx.classify_xyz(*eight)
        assert np.array_equal(lg, lb8)


def test_storage_order_invariance(ctx_big):
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("cfg2", 71)
    perm = np.arange(64 * 2048).reshape(2048, 64).T.reshape(-1)
    ctx_big.set_params(p)
    l1, _ = ctx_big.classify_xyz(x, y, z)
    l2, _ = ctx_big.classify_xyz(x[perm], y[perm], z[perm])
    assert np.array_equal(l1[perm], l2)


@pytest.mark.parametrize("cfg,seed", [("cfg2", 5), ("narrow", 6), ("default_roi", 7), ("cfg5", 3), ("cfg1", 2)])
def test_published_order(ctx_big, cfg, seed):
    """urf_ordered_indices: the road / curb / road_probably clouds as the reference publishes them
    (ring by ring, ascending azimuth -- its per-ring quicksort).  cfg5 has 4096-point rings, which
    take the global-memory sort path."""
    p = O.cfg_params(cfg)
    x, y, z = O.cfg_cloud(cfg, seed)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    road, curb, r10 = ctx_big.ordered_indices(len(x))
    assert np.array_equal(road, st["road_order"])
    assert np.array_equal(curb, st["curb_order"])
    assert np.array_equal(r10, st["ring10_order"])
    # and they are the same sets the label bytes describe
    assert np.array_equal(np.sort(road), np.nonzero((lg & 3) == 1)[0]) and np.array_equal(np.sort(curb), np.nonzero((lg & 3) == 2)[0])


def test_index_lists(ctx_big):
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("cfg2", 81)
    n = len(x)
    ctx_big.set_params(p)
    lg, _ = ctx_big.classify_xyz(x, y, z)
    dl = DevBuf.from_numpy(lg)
    bufs = [DevBuf(4 * n) for _ in range(4)]
    dc = DevBuf(16)
    ctx_big.compact_indices(dl, n, *bufs, dc)
    ctx_big.synchronize()
    cnt = dc.to_numpy(np.uint32)
    want = [np.nonzero((lg & 3) == 1)[0], np.nonzero((lg & 3) == 2)[0], np.nonzero(lg & 4)[0], np.nonzero(lg & 16)[0]]
    for k in range(4):
        assert cnt[k] == len(want[k])
        assert np.array_equal(bufs[k].to_numpy(np.uint32, int(cnt[k])), want[k])


def test_batch_outputs_equal_the_single_scan_outputs():
    """the index sets, the published order and the marker points of EVERY scan of a batch
    in one launch sequence each (grid over tiles | rings x scans) -- against numpy on the label bytes
    and against the single-scan entry points."""
    p = O.cfg_params("cfg2")
    n = 20000   # not a multiple of the 2048-label tile
    scans = [tuple(a[:n].copy() for a in O.cfg_cloud("narrow" if k % 2 else "cfg2", 60 + k)) for k in range(5)]
    S = len(scans)
    with u.Context(n, S, params=p) as ctx:
        X, Y, Z = (np.concatenate([s[k] for s in scans]) for k in range(3))
        dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
        dl = DevBuf(S * n)
        ctx.classify_batch_soa(dx, dy, dz, n, S, dl, None)
        ctx.synchronize()
        L = dl.to_numpy(np.uint8).reshape(S, n)
        # index sets
        bufs = [DevBuf(4 * S * n) for _ in range(4)]
        dc = DevBuf(16 * S)
        ctx.compact_indices_batch(dl, n, S, *bufs, dc)
        ctx.synchronize()
        cnt = dc.to_numpy(np.uint32).reshape(S, 4)
        lists = [b.to_numpy(np.uint32).reshape(S, n) for b in bufs]
        for s in range(S):
            want = [np.nonzero((L[s] & 3) == 1)[0], np.nonzero((L[s] & 3) == 2)[0], np.nonzero(L[s] & 4)[0], np.nonzero(L[s] & 16)[0]]
            for k in range(4):
                assert cnt[s, k] == len(want[k]) and np.array_equal(lists[k][s, :cnt[s, k]], want[k]), (s, k)
        # published order and marker points: batch == scan by scan
        ob = [DevBuf(4 * S * n) for _ in range(3)]
        oc = DevBuf(12 * S)
        ctx.ordered_indices_batch(*ob, n, oc)
        mp, mc = DevBuf(4 * S * 361 * 4), DevBuf(4 * S)
        ctx.marker_points_batch(mp, mc)
        ctx.synchronize()
        ocnt = oc.to_numpy(np.uint32).reshape(S, 3)
        olists = [b.to_numpy(np.uint32).reshape(S, n) for b in ob]
        mpts, mcnt = mp.to_numpy(np.float32).reshape(S, 361, 4), mc.to_numpy(np.uint32)
        for s in range(S):
            single = ctx.ordered_indices(n, scan=s)
            for k in range(3):
                assert np.array_equal(olists[k][s, :ocnt[s, k]], single[k]), (s, k)
            lb, ib, st = O.run_b(*scans[s], p, debug=True)
            assert np.array_equal(single[0], st["road_order"]) and np.array_equal(single[1], st["curb_order"])
            assert np.array_equal(mpts[s, :mcnt[s]], ctx.marker_points(scan=s))


def test_full_size_batch_properties():
    """BASELINE cfg3 size (1024 scans of 64x2048): 64 distinct sweeps repeated 16 times.
    Properties: every copy of a sweep gets identical labels wherever it sits in the batch;
    a second pass over the batch is idempotent; the 64 distinct results equal oracle B; the
    per-scan counters add up to the label histogram."""
    S, R = 1024, 64
    p = O.cfg_params("cfg2")
    uniq = [O.cfg_cloud("cfg2" if s % 4 else "narrow", 100 + s) for s in range(R)]
    n = 64 * 2048
    X = np.concatenate([c[0] for c in uniq])
    Y = np.concatenate([c[1] for c in uniq])
    Z = np.concatenate([c[2] for c in uniq])
    reps = S // R
    dx, dy, dz = (DevBuf.from_numpy(np.tile(a, reps)) for a in (X, Y, Z))
    dl, di = DevBuf(S * n), DevBuf(32 * S)
    with u.Context(n, S, params=p) as ctx:
        ctx.classify_batch_soa(dx, dy, dz, n, S, dl, di)
        L1 = dl.to_numpy(np.uint8).reshape(S, n)
        I1 = di.to_numpy(np.uint32).reshape(S, 8)
        ctx.classify_batch_soa(dx, dy, dz, n, S, dl, di)
        L2 = dl.to_numpy(np.uint8).reshape(S, n)
    assert np.array_equal(L1, L2)
    for r in range(1, reps):
        assert np.array_equal(L1[r * R:(r + 1) * R], L1[:R])
        assert np.array_equal(I1[r * R:(r + 1) * R], I1[:R])
    assert np.array_equal((L1[:R] & 3 == 1).sum(1), I1[:R, 4]) and np.array_equal((L1[:R] & 3 == 2).sum(1), I1[:R, 5])
    for s in range(R):
        lb, ib, _ = O.run_b(*uniq[s], p)
        assert np.array_equal(L1[s], lb), "scan %d" % s


def test_device_arithmetic_selftest(ctx_hooks):
    ctx_big = ctx_hooks
    """The kernels divide by pi with three fma-class operations instead of an f64 division; that
    must equal the IEEE quotient for EVERY float the path can produce (exhaustive over [0, 600])."""
    assert ctx_big.selftest() == 0


def test_fast_path_error_bounds(ctx_hooks):
    ctx_big = ctx_hooks
    """Ring and sector are decided from float approximations of the angles wherever the
    approximation is clear of every decision boundary by a margin (urf_device.hpp); the margins
    (3e-4 deg, 2e-6 rad, 2.5e-4, the azimuth's 5e-4 + 6e-4 / delta deg) must dominate the error measured on 2^28 pseudo-random points."""
    ctx_big.set_params(u.default_params())
    ev, ea, eu, ez = ctx_big.selftest_fast(1 << 28)
    assert ev < 1.0e-4 and ea < 0.7e-6 and eu < 0.85e-4 and ez < 0.5, (ev, ea, eu, ez)   # ez: fraction of the azimuth margin
    # the sector margin scales with the number of sectors (urf_dev_params::sector_margin): at the
    # largest supported count the measured error of the scaled polar angle must stay below a third of it
    p = u.default_params()
    p.sectors = 1022
    ctx_big.set_params(p)
    _, _, eu, _ = ctx_big.selftest_fast(1 << 26)
    assert eu < 2.5e-4 * (1022 / 360) / 3, eu
    ctx_big.set_params(u.default_params())


@pytest.mark.parametrize("log2_scale", [0, -30, 30, -62])
@pytest.mark.parametrize("tweak", [{}, {"starbeam_filter": 1, "xDirection": 1}, {"channels": 20}])
def test_decision_boundaries(ctx_big, log2_scale, tweak):
    """Ring, sector and window membership are decided on float approximations wherever those are
    clear of the boundary by a proven margin, and on the reference's exact sequence inside it.
    This cloud sits on the boundaries (both sides of every margin), at the sensor's scale and at
    2^-30, 2^30 and 2^-62 times it (the last one outside the range the fast paths accept)."""
    sc = 2.0 ** log2_scale
    x, y, z = O.boundary_cloud(sc)
    p = u.default_params()
    for k, v in tweak.items():
        setattr(p, k, v)
    p.min_X, p.max_X, p.min_Y, p.max_Y, p.min_Z, p.max_Z = -60 * sc, 60 * sc, -60 * sc, 60 * sc, -3 * sc, -1 * sc
    p.channels = tweak.get("channels", 32)   # 20: the table fills up, later unmatched points stay without a ring
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    assert ib["status"] == 0 and ib["n_rings"] >= 14
    ctx_big.set_params(p)
    n = len(x)
    ctx_big.enable_stage_capture(2)   # the production decisions, ring / sector per input point recorded
    try:
        lg, ig = ctx_big.classify_xyz(x, y, z)
        assert np.array_equal(ctx_big.read_stage(u.STAGE_ANGLE_TABLE, n), st["angle_table"])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_RING, n), st["ring"])
        assert np.array_equal(ctx_big.read_stage(u.STAGE_SECTOR, n), st["sector"])
        assert np.array_equal(lg, lb) and info_equal(ig, ib)
    finally:
        ctx_big.enable_stage_capture(0)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, n), st["detect"])
    assert np.array_equal(ctx_big.read_stage(u.STAGE_BEAM_STOP, n), st["beam_stop"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_rough_ground_fills_the_candidate_list(ctx_big):
    """k_ring collects the points that need a detector's angle test in a per-ring list and works it
    off when it might overflow: on rough ground (noise above curbHeight) nearly every point is a
    candidate, so the list is flushed in the middle of the rings."""
    x, y, z = O.cfg_cloud("cfg2", 9)
    rng = np.random.default_rng(9)
    z = (z + 0.03 * rng.standard_normal(len(z))).astype(np.float32)
    p = O.cfg_params("cfg2")
    p.curbHeight = 0.01
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    assert ib["n_curb"] > 2000                      # the detectors fire all over the place
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(ctx_big.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_whole_tiles_on_integer_degrees(ctx_big):
    """k_label decides window membership on the approximate azimuth unless it is within the margin of
    an integer degree or a window end; such points are listed (256 per tile) and decided on the
    exact value, the overflow in place.  Here every point sits on an integer degree."""
    rng = np.random.default_rng(11)
    n = 6000
    fi = rng.choice([31.0, 32.0, 60.0, 61.0, 119.0, 200.0, 271.0, 330.0], n)
    va = 62.0 + 1.7 * rng.integers(0, 14, n)       # 14 rings
    rho = 1.8 * np.tan(np.deg2rad(va)) * (1.0 + 1e-3 * np.arange(n) / n)
    x = (rho * np.cos(np.deg2rad(fi - 90.0))).astype(np.float32)   # azimuth: 0 at -y, 90 at +x
    y = (rho * np.sin(np.deg2rad(fi - 90.0))).astype(np.float32)
    z = np.full(n, -1.8, np.float32)
    z[rng.integers(0, n, 40)] += 0.3               # a few curb stones so that beams get blocked
    p = u.default_params()
    p.min_X, p.max_X, p.min_Y, p.max_Y = -60, 60, -60, 60
    p.channels = 32
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    az = st["azimuth"][st["ring"] >= 0]
    assert (np.abs(az - np.round(az)) < 5e-4).mean() > 0.9 and ib["n_road"] > 100
    ctx_big.set_params(p)
    lg, ig = ctx_big.classify_xyz(x, y, z)
    assert np.array_equal(lg, lb) and info_equal(ig, ib)


def test_capacity_and_argument_errors():
    with u.Context(4096, 2) as ctx:
        x, y, z = [a[:8192] for a in O.cfg_cloud("cfg2", 1)]
        with pytest.raises(u.UrfError) as e:
            ctx.classify_xyz(x, y, z)
        assert e.value.code == -4
        p = u.default_params()
        p.channels = 500
        with pytest.raises(u.UrfError) as e:
            ctx.set_params(p)
        assert e.value.code == -6
        p = u.default_params()
        p.size = 8
        with pytest.raises(u.UrfError):
            ctx.set_params(p)


@pytest.mark.parametrize("seed", range(24))
def test_published_order_and_markers_with_equal_azimuths(seed):
    """Clouds whose x / y lie on a grid: many points of a ring share their azimuth to the bit (and their planar range).  The
    published order and the marker points then depend on the order the reference's Lomuto quicksort leaves equal azimuths in
    (lidar_segmentation.cpp:70-93; deterministic): k_ring_order / k_marker_ring run it literally for such a ring."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([300, 3000, 20000]))
    cloud = fuzz.random_cloud(rng, n, tie_grid=float(rng.choice([1 / 16, 1 / 4, 1.0])))
    p = fuzz.random_params(rng)
    p.channels = int(rng.choice([16, 64]))
    lb, ib, st = O.run_b(*cloud, p, debug=True)
    with u.Context(max(len(cloud[0]), 64), 1, params=p) as ctx:
        lg, ig = ctx.classify_xyz(*cloud)
        assert np.array_equal(lg, lb)
        if ib["status"] == 0:
            road, curb, prob = ctx.ordered_indices(len(cloud[0]))
            assert np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"])
            assert np.array_equal(prob, st["ring10_order"])
            mg = ctx.marker_points()
            assert mg.shape == st["marker_pts"].shape and np.array_equal(mg, st["marker_pts"])
