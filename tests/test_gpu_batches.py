"""Multi-scan batches of the non-headline shapes (missing #4): everything that is
indexed by scan inside one launch sequence -- the persistent work lists of oversized star sectors,
the general k_ring instance, the speculative ring table and its repair, the empty-tile shortcuts for a
region of interest that drops whole azimuth ranges -- checked against oracle B scan by scan."""
import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u
from hipmem import DevBuf
from test_gpu_parity import check_against_b, crowded_cloud, run_batch

pytestmark = pytest.mark.gpu
N = 64 * 2048


def rolled(cloud, cols, rings=64):
    """The same sweep stored from another start column (firing order kept)."""
    return tuple(np.roll(a.reshape(-1, rings), cols, axis=0).reshape(-1).copy() for a in cloud)


def test_cfg5_batch_with_mid_and_big_sectors():
    """128 x 4096 sweeps (every sector ~1456 points: the mid work list spans the scans of the batch), one of
    them with 6000 extra points in two sectors (> 2048 points: the big list) and one stored in another
    order; ragged batch, channels 128 / interval 0.05 (BASELINE cfg5, n_scans > 1)."""
    p = O.cfg_params("cfg5")
    a = O.cfg_cloud("cfg5", 21)
    b = O.cfg_cloud("cfg5", 22)
    xc, yc, zc = crowded_cloud(6000, seed=23)
    c = tuple(np.concatenate([v, w]) for v, w in zip(O.cfg_cloud("cfg5", 24), (xc, yc, zc)))
    d = rolled(O.cfg_cloud("cfg5", 25), 1000, rings=128)
    scans = [a, c, b, d]
    with u.Context(max(len(s[0]) for s in scans), len(scans)) as ctx:
        labels, infos = run_batch(ctx, scans, p, ragged=True)
        check_against_b(labels, infos, scans, p)
        labels, infos = run_batch(ctx, scans[::-1], p, ragged=True)   # the same scans at other batch positions
        check_against_b(labels, infos, scans[::-1], p)


@pytest.mark.parametrize("tweak", [{"curbPoints": 9, "beamZone": 45.5}, {"curbPoints": 2}, {"curbPoints": 30, "curbHeight": 0.03}])
def test_general_ring_kernel_in_a_batch(tweak):
    """curbPoints != 5 takes k_ring_general (x / y / z windows in LDS, exact azimuth only for curb points)."""
    p = O.cfg_params("cfg2")
    for k, v in tweak.items():
        setattr(p, k, v)
    scans = [O.cfg_cloud("cfg2" if s % 2 else "narrow", 300 + s) for s in range(5)]
    scans[3] = rolled(scans[3], 777)
    with u.Context(N, len(scans)) as ctx:
        labels, infos = run_batch(ctx, scans, p)
        check_against_b(labels, infos, scans, p)


def test_default_roi_batch_where_one_scan_defeats_the_speculation():
    """The reference's default region of interest (cfg/LidarFilters.cfg:42-51) keeps a wedge in front of the
    sensor.  A sweep stored from the rear has no ROI point among its first 8192: k_ring_table gives up with an
    empty table, k_split notices, the scan is repaired inside the same call -- scan 1 of 3 only -- and the
    context stops speculating.  Labels equal oracle B before and after."""
    p = O.cfg_params("default_roi")
    front = [O.cfg_cloud("default_roi", 40 + s) for s in range(2)]
    rear = rolled(O.cfg_cloud("default_roi", 42), 1024)
    scans = [front[0], rear, front[1]]
    with u.Context(N, 3) as ctx:
        for rep in range(2):
            labels, infos = run_batch(ctx, scans, p)
            check_against_b(labels, infos, scans, p)
            assert infos[1][2] == infos[0][2] == infos[2][2]      # the same rings either way
    # the other way round: a context that never has to repair anything
    with u.Context(N, 3) as ctx:
        labels, infos = run_batch(ctx, [front[0], front[1], front[0]], p)
        check_against_b(labels, infos, [front[0], front[1], front[0]], p)


def test_full_size_batch_default_roi():
    """BASELINE cfg3 size (1024 scans of 64x2048) with the reference's DEFAULT region of interest: more than
    half of the 2048-point tiles hold no ROI point (k_split / k_label leave early there), rings are cut into
    runs of unequal length (k_ring's run table).  64 distinct sweeps -- a quarter of them stored from another
    start column, so that the empty tiles sit elsewhere -- repeated 16 times: every copy gets identical labels
    wherever it sits, a second pass is idempotent, all 64 equal oracle B, counters match the labels."""
    S, R = 1024, 64
    p = O.cfg_params("default_roi")
    uniq = []
    for s in range(R):
        c = O.cfg_cloud("default_roi" if s % 3 else "narrow", 500 + s)
        uniq.append(rolled(c, 37 * s) if s % 4 == 1 else c)
    X = np.concatenate([c[0] for c in uniq])
    Y = np.concatenate([c[1] for c in uniq])
    Z = np.concatenate([c[2] for c in uniq])
    reps = S // R
    dx, dy, dz = (DevBuf.from_numpy(np.tile(a, reps)) for a in (X, Y, Z))
    dl, di = DevBuf(S * N), DevBuf(32 * S)
    dl.fill(0xEE)
    with u.Context(N, S, params=p) as ctx:
        ctx.classify_batch_soa(dx, dy, dz, N, S, dl, di)
        L1 = dl.to_numpy(np.uint8).reshape(S, N)
        I1 = di.to_numpy(np.uint32).reshape(S, 8)
        ctx.classify_batch_soa(dx, dy, dz, N, S, dl, di)
        L2 = dl.to_numpy(np.uint8).reshape(S, N)
    for b in (dx, dy, dz, dl, di):
        b.free()
    assert np.array_equal(L1, L2)
    for r in range(1, reps):
        assert np.array_equal(L1[r * R:(r + 1) * R], L1[:R])
        assert np.array_equal(I1[r * R:(r + 1) * R], I1[:R])
    assert np.array_equal((L1[:R] & 3 == 1).sum(1), I1[:R, 4]) and np.array_equal((L1[:R] & 3 == 2).sum(1), I1[:R, 5])
    assert np.array_equal(((L1[:R] & u.FLAG_ROI) != 0).sum(1), I1[:R, 1])
    for s in range(R):
        lb, ib, _ = O.run_b(*uniq[s], p)
        assert np.array_equal(L1[s], lb), "scan %d" % s
        assert I1[s, 7] == 0
