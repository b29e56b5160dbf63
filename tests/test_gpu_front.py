"""The fused front end for batches of sweeps in firing order (urban_road_filter_amd/csrc/urf_front.hpp: k_front, k_front_finish,
k_label_front) through the C ABI: labels and summaries against oracle B -- on sweeps that take it (analytic and sensor-like, with
drop-outs, cut by the reference's default region of interest, lasers in any fixed order inside a firing, partial last tiles), on
sweeps that must NOT take it (stored from another column, shuffled, points on the sensor's axis) next to ones that do, through the
list-driven and the full-grid legacy kernels; what the context does around it (urf_set_front_mode, urf_front_scans, the entry
points that read ring-sorted intermediate results)."""
import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u
from fuzz_organised import case
from hipmem import DevBuf
from test_gpu_parity import check_against_b, run_batch

pytestmark = pytest.mark.gpu
N = 64 * 2048


def fused_batch(ctx, scans, p, mode=2, ragged=False):
    ctx.set_front_mode(mode)
    labels, infos = run_batch(ctx, scans, p, ragged=ragged)
    return labels, infos, ctx.front_scans()


def permuted(cloud, perm):
    """The lasers of every firing in another (fixed) order: what a driver that reports in laser-number order delivers."""
    return tuple(np.ascontiguousarray(a.reshape(-1, 64)[:, perm].reshape(-1)) for a in cloud)


def rolled(cloud, cols):
    return tuple(np.ascontiguousarray(np.roll(a.reshape(-1, 64), cols, axis=0).reshape(-1)) for a in cloud)


@pytest.mark.parametrize("name,seeds", [("cfg2", (1, 2, 3, 4)), ("narrow", (1, 2, 3)), ("sensor", (1, 2, 3, 4)), ("sensor_narrow", (1, 2)),
                                        ("default_roi", (1, 2, 3)), ("sensor_default_roi", (1, 2))])
def test_sweeps_in_firing_order_take_the_fused_front_end(name, seeds):
    p = O.cfg_params(name)
    scans = [O.cfg_cloud(name, s) for s in seeds]
    with u.Context(N, len(scans)) as ctx:
        labels, infos, nf = fused_batch(ctx, scans, p)
        assert nf == len(scans)
        check_against_b(labels, infos, scans, p)
        labels, infos, nf = fused_batch(ctx, scans[::-1], p)   # other batch positions, the row's previous ring count as a hint
        assert nf == len(scans)
        check_against_b(labels, infos, scans[::-1], p)


def test_lasers_in_any_fixed_order():
    """Lane l of a firing is laser l, whatever table entry that laser sits on: learned from the lane's first point."""
    p = O.cfg_params("sensor")
    perm = np.random.default_rng(5).permutation(64)
    scans = [permuted(O.cfg_cloud("sensor", s), perm) for s in (1, 2)] + [permuted(O.cfg_cloud("cfg2", 3), perm[::-1].copy())]
    with u.Context(N, len(scans)) as ctx:
        labels, infos, nf = fused_batch(ctx, scans, p)
        assert nf == len(scans)
        check_against_b(labels, infos, scans, p)


@pytest.mark.parametrize("tweak", [{"xDirection": 1}, {"xDirection": 2}, {"starbeam_filter": 1}, {"star_shaped_method": 0},
                                   {"x_zero_method": 0}, {"z_zero_method": 0}, {"blind_spots": 0}, {"curbHeight": 0.01},
                                   {"interval": 0.1, "angleFilter1": 120.0, "angleFilter2": 100.0}])
def test_parameters(tweak):
    """(curbHeight 0.01: rings with more curb points than their list holds -- the per-degree tables of k_front_finish)"""
    p = O.cfg_params("cfg2")
    for k, v in tweak.items():
        setattr(p, k, v)
    scans = [O.cfg_cloud("narrow", 1), O.cfg_cloud("sensor", 2), O.cfg_cloud("cfg2", 3)]
    with u.Context(N, len(scans)) as ctx:
        labels, infos, nf = fused_batch(ctx, scans, p)
        assert nf == len(scans)
        check_against_b(labels, infos, scans, p)


def test_scans_without_the_shape_are_handed_back():
    """A shuffled sweep and one stored from another column (its sectors fall inside a tile) between organised ones: the legacy
    kernels take them in the same call -- list-driven on a context that has not seen such a scan, as full grids afterwards."""
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("cfg2", 3)
    pm = np.random.default_rng(1).permutation(len(x))
    scans = [O.cfg_cloud("cfg2", 1), (x[pm], y[pm], z[pm]), O.cfg_cloud("sensor", 2), rolled(O.cfg_cloud("narrow", 4), 700)]
    with u.Context(N, len(scans)) as ctx:
        for _ in range(3):   # first call: lists; then grids
            labels, infos, nf = fused_batch(ctx, scans, p)
            assert nf == 2
            check_against_b(labels, infos, scans, p)
        labels, infos, nf = fused_batch(ctx, scans, p, mode=0)
        assert nf == 0
        check_against_b(labels, infos, scans, p)


def test_a_batch_of_unorganised_clouds_switches_it_off():
    """Every scan handed back: the context stops trying (mode 1), urf_set_front_mode starts over."""
    p = O.cfg_params("cfg2")
    rng = np.random.default_rng(3)
    scans = []
    for s in range(3):
        x, y, z = O.cfg_cloud("cfg2", 10 + s)
        pm = rng.permutation(len(x))
        scans.append((x[pm], y[pm], z[pm]))
    good = [O.cfg_cloud("cfg2", 20 + s) for s in range(3)]
    with u.Context(N, 3) as ctx:
        ctx.set_front_mode(2)
        labels, infos, nf = fused_batch(ctx, scans, p)
        assert nf == 0
        check_against_b(labels, infos, scans, p)
        ctx.set_front_mode(1)
        labels, infos = run_batch(ctx, good, p)          # (mode 1: fewer than 192 scans never take it anyway)
        assert ctx.front_scans() == 0
        check_against_b(labels, infos, good, p)
        labels, infos, nf = fused_batch(ctx, good, p)    # mode 2 again: a new start
        assert nf == 3
        check_against_b(labels, infos, good, p)


def test_rear_stored_default_roi_sweeps_repair_their_ring_table():
    """The speculative ring table of a sweep stored from the rear is incomplete; k_front notices like k_split does, the scan is
    repaired and split the legacy way in the same call."""
    p = O.cfg_params("default_roi")
    scans = [rolled(O.cfg_cloud("default_roi", s), 1024) for s in (1, 2)] + [O.cfg_cloud("default_roi", 3)]
    for _ in range(2):
        with u.Context(N, len(scans)) as ctx:
            labels, infos, nf = fused_batch(ctx, scans, p)
            assert nf == 1
            check_against_b(labels, infos, scans, p)
            labels, infos, nf = fused_batch(ctx, scans, p)   # (the context has stopped speculating on the ring table)
            check_against_b(labels, infos, scans, p)


@pytest.mark.parametrize("cols", [96, 40, 33, 2047])
def test_partial_last_tile_and_ragged_batches(cols):
    p = O.cfg_params("cfg2")
    a = u.synth_cloud(64, cols, 1, 7)
    b = u.synth_cloud(64, 512, 3, 8)
    scans = [a, b, tuple(v[:64 * 300 + 17].copy() for v in b)]   # (the last one ends inside a firing)
    with u.Context(64 * 2048, len(scans)) as ctx:
        labels, infos, nf = fused_batch(ctx, scans, p, ragged=True)
        assert nf >= 2
        check_against_b(labels, infos, scans, p)


def test_points_on_the_axis_and_too_few_points():
    """A ring point with x == y == 0 (NaN azimuth: k_nan_rings, legacy kernels) and a scan below the 30-point threshold."""
    p = O.cfg_params("cfg2")
    a = tuple(v.copy() for v in O.cfg_cloud("cfg2", 5))
    a[0][64 * 100 + 7] = 0.0
    a[1][64 * 100 + 7] = 0.0
    few = tuple(v.copy() for v in O.cfg_cloud("cfg2", 6))
    few[0][29:] = 1.0e6
    scans = [a, few, O.cfg_cloud("sensor", 7)]
    with u.Context(N, len(scans)) as ctx:
        ctx.set_front_mode(2)
        labels, infos = run_batch(ctx, scans, p)
        for k, (x, y, z) in enumerate(scans):
            lb, ib, _ = O.run_b(x, y, z, p)
            assert np.array_equal(labels[k], lb), k
            assert int(np.int32(infos[k][0])) == ib["status"] and all(int(infos[k][j]) == ib[f] for j, f in
                                                                       ((1, "n_roi"), (3, "n_ring_pts"), (4, "n_road"), (5, "n_curb"), (6, "n_ring10"),
                                                                        (7, "n_nan_azimuth"))), k
        assert ctx.front_scans() == 2   # (the too-few scan keeps its flag: nothing is published for it either way)


@pytest.mark.parametrize("seed", range(40))
def test_organised_sweeps_with_holes(seed):
    """tests/fuzz_organised.py through the fused front end (curbPoints != 5 takes the legacy kernels)."""
    (x, y, z), p = case(7_300_000 + seed)
    lb, ib, _ = O.run_b(x, y, z, p)
    with u.Context(len(x), 1) as ctx:
        labels, infos, nf = fused_batch(ctx, [(x, y, z)], p)
    assert np.array_equal(labels[0], lb), "%d labels differ (fused %d)" % (int((labels[0] != lb).sum()), nf)
    keys = ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10")
    assert {f: int(v) for f, v in zip(keys, infos[0][:7])} == {f: ib[f] for f in keys}
    assert p.curbPoints == 5 or nf == 0


def test_ring_sorted_results_after_a_fused_call():
    """urf_ordered_indices / urf_marker_points / urf_read_stage read ring-sorted intermediate results: the call is run again
    through the legacy kernels, the context stays with them."""
    p = O.cfg_params("cfg2")
    scans = [O.cfg_cloud("sensor", 1), O.cfg_cloud("narrow", 2)]
    with u.Context(N, 2) as ctx:
        X, Y, Z = (np.concatenate([s[k] for s in scans]) for k in range(3))
        dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
        dl = DevBuf(2 * N)
        ctx.set_params(p)
        ctx.set_front_mode(2)
        ctx.classify_batch_soa(dx, dy, dz, N, 2, dl, None)
        assert ctx.front_scans() == 2
        for k, (x, y, z) in enumerate(scans):
            lb, ib, st = O.run_b(x, y, z, p, debug=True)
            road, curb, prob = ctx.ordered_indices(N, scan=k)
            assert np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"]) and np.array_equal(prob, st["ring10_order"])
            assert np.array_equal(ctx.marker_points(scan=k), st["marker_pts"])
            assert np.array_equal(ctx.read_stage(u.STAGE_DETECT, N, scan=k), st["detect"])
            assert np.array_equal(dl.to_numpy(np.uint8).reshape(2, N)[k], lb)
        assert ctx.front_scans() == 0                      # (the call was run again through the legacy kernels)
        ctx.set_front_mode(1)
        ctx.classify_batch_soa(dx, dy, dz, N, 2, dl, None)
        assert ctx.front_scans() == 0                      # ... and the context stays with them
        ctx.set_front_mode(2)
        ctx.classify_batch_soa(dx, dy, dz, N, 2, dl, None)
        assert ctx.front_scans() == 2


def test_stage_capture_and_other_shapes_keep_the_legacy_kernels():
    p = O.cfg_params("cfg2")
    scans = [O.cfg_cloud("cfg2", 1)]
    with u.Context(N, 1) as ctx:
        ctx.enable_stage_capture(2)
        labels, infos, nf = fused_batch(ctx, scans, p)
        assert nf == 0
        check_against_b(labels, infos, scans, p)
        ctx.enable_stage_capture(0)
        p9 = O.cfg_params("cfg2")
        p9.curbPoints = 9
        labels, infos, nf = fused_batch(ctx, scans, p9)
        assert nf == 0
        check_against_b(labels, infos, scans, p9)
    p5 = O.cfg_params("cfg5")
    with u.Context(128 * 1024, 1) as ctx:
        c5 = u.synth_cloud(128, 1024, 1, 3)
        labels, infos, nf = fused_batch(ctx, [c5], p5)
        assert nf == 0
        check_against_b(labels, infos, [c5], p5)


def test_mode_one_takes_batches_of_192_scans():
    """urf_set_front_mode(1), the default: the fused kernels from URF_FRONT_MIN_SCANS = 192 scans per call on (below that the general
    kernels are faster: tools/r6_min_scans.py); two tiles per block of k_front up to 511 scans, four from 512 on."""
    p = O.cfg_params("cfg2")
    base = [u.synth_cloud(64, 256, 1 + (s % 2) * 2, 50 + s) for s in range(8)]
    scans = [base[s % 8] for s in range(200)]
    with u.Context(64 * 256, 520) as ctx:
        ctx.set_front_mode(1)
        labels, infos = run_batch(ctx, scans, p)
        assert ctx.front_scans() == 200
        check_against_b(labels[:8], infos[:8], scans[:8], p)
        for s in range(8, 200):
            assert np.array_equal(labels[s], labels[s % 8]) and np.array_equal(infos[s], infos[s % 8])
        labels, infos = run_batch(ctx, scans[:191], p)
        assert ctx.front_scans() == 0
        check_against_b(labels[:8], infos[:8], scans[:8], p)
        many = [base[s % 8] for s in range(520)]                # (four tiles per block)
        labels, infos = run_batch(ctx, many, p)
        assert ctx.front_scans() == 520
        for s in range(520):
            assert np.array_equal(labels[s], labels[s % 8]) and np.array_equal(infos[s], infos[s % 8])
        check_against_b(labels[:8], infos[:8], many[:8], p)


def ring_major(cloud):
    """The same sweep stored ring by ring (row-major 64 x W: what an Ouster's driver delivers as an organised cloud)."""
    return tuple(np.ascontiguousarray(a.reshape(-1, 64).T.reshape(-1)) for a in cloud)


def shuffled(cloud, seed):
    pm = np.random.default_rng(seed).permutation(len(cloud[0]))
    return tuple(a[pm].copy() for a in cloud)


@pytest.mark.parametrize("mode", [2, 0])
def test_storage_orders_of_sensor_like_sweeps_against_the_oracle(mode):
    """The reference makes no assumption about the order its input arrives in (lidar_segmentation.cpp:100-117, 221-278), and
    with equal planar ranges in a star sector the labels DEPEND on it (std::sort's tie order): sensor-like sweeps -- ties in
    every sector -- ring-major, in laser order and shuffled, each against oracle B on the same input.  Ring-major: every
    sector is 64 runs of ~6 points (k_star_sort_runs)."""
    p = O.cfg_params("sensor")
    perm = np.random.default_rng(9).permutation(64)
    scans = [ring_major(O.cfg_cloud("sensor", 1)), permuted(O.cfg_cloud("sensor", 2), perm), shuffled(O.cfg_cloud("sensor", 3), 4),
             ring_major(O.cfg_cloud("sensor_narrow", 5)), O.cfg_cloud("sensor", 6)]
    with u.Context(N, len(scans)) as ctx:
        labels, infos, nf = fused_batch(ctx, scans, p, mode=mode)
        assert nf == (2 if mode else 0)
        check_against_b(labels, infos, scans, p)
    pr = O.cfg_params("sensor_default_roi")
    scans = [ring_major(O.cfg_cloud("sensor_default_roi", 7)), shuffled(O.cfg_cloud("sensor_default_roi", 8), 9)]
    with u.Context(N, len(scans)) as ctx:
        labels, infos, nf = fused_batch(ctx, scans, pr, mode=mode)
        check_against_b(labels, infos, scans, pr)


# ---- row-major organised sweeps (height = the 64 lasers, width = firings): k_rows_probe, k_transpose, k_label_front's row-major stores ----
def rows_then_fused(ctx, scans, p, want=None, ragged=False):
    """A context's first call with such sweeps only sights the layout (legacy kernels, labels already the oracle's); from the second on
    they take k_transpose and the fused kernels."""
    labels, infos, nf0 = fused_batch(ctx, scans, p, ragged=ragged)
    check_against_b(labels, infos, scans, p)
    labels, infos, nf = fused_batch(ctx, scans, p, ragged=ragged)
    check_against_b(labels, infos, scans, p)
    if want is not None:
        assert nf == want, (nf0, nf)
    return nf0, nf


@pytest.mark.parametrize("name,seeds", [("cfg2", (1, 2, 3, 4)), ("narrow", (1, 2, 3)), ("sensor", (1, 2, 3, 4)), ("sensor_narrow", (1, 2)),
                                        ("default_roi", (1, 2, 3)), ("sensor_default_roi", (1, 2))])
def test_row_major_sweeps_take_the_fused_front_end_from_the_second_call(name, seeds):
    p = O.cfg_params(name)
    scans = [ring_major(O.cfg_cloud(name, s)) for s in seeds]
    with u.Context(N, len(scans)) as ctx:
        nf0, nf = rows_then_fused(ctx, scans, p, want=len(scans))
        assert nf0 == 0
        labels, infos, nf = fused_batch(ctx, scans[::-1], p)
        assert nf == len(scans)
        check_against_b(labels, infos, scans[::-1], p)
        labels, infos, nf = fused_batch(ctx, scans, p, mode=0)   # ... and the legacy kernels on the same context
        assert nf == 0
        check_against_b(labels, infos, scans, p)


def test_row_major_lasers_in_any_order_rear_stored_and_mixed_batches():
    p = O.cfg_params("sensor")
    perm = np.random.default_rng(5).permutation(64)
    scans = [ring_major(permuted(O.cfg_cloud("sensor", s), perm)) for s in (1, 2)] + [ring_major(permuted(O.cfg_cloud("cfg2", 3), perm[::-1].copy()))]
    with u.Context(N, len(scans)) as ctx:
        rows_then_fused(ctx, scans, p, want=len(scans))
    # every row starts outside the reference's default region of interest (the sweep is stored from the rear)
    pr = O.cfg_params("default_roi")
    scans = [ring_major(rolled(O.cfg_cloud("default_roi", s), 1024)) for s in (1, 2)] + [ring_major(O.cfg_cloud("default_roi", 3))]
    with u.Context(N, len(scans)) as ctx:
        rows_then_fused(ctx, scans, pr, want=len(scans))
    # row-major, shuffled, firing order and row-major again in one batch
    p2 = O.cfg_params("cfg2")
    scans = [ring_major(O.cfg_cloud("cfg2", 1)), shuffled(O.cfg_cloud("cfg2", 3), 1), O.cfg_cloud("sensor", 2), ring_major(O.cfg_cloud("sensor", 4))]
    with u.Context(N, len(scans)) as ctx:
        nf0, nf = rows_then_fused(ctx, scans, p2, want=3)
        assert nf0 == 1


@pytest.mark.parametrize("tweak", [{"xDirection": 1}, {"starbeam_filter": 1}, {"star_shaped_method": 0}, {"curbHeight": 0.01},
                                   {"x_zero_method": 0}, {"z_zero_method": 0}, {"blind_spots": 0}])
def test_row_major_parameters(tweak):
    p = O.cfg_params("cfg2")
    for k, v in tweak.items():
        setattr(p, k, v)
    scans = [ring_major(O.cfg_cloud("sensor", 1)), ring_major(O.cfg_cloud("narrow", 2))]
    with u.Context(N, len(scans)) as ctx:
        rows_then_fused(ctx, scans, p, want=len(scans))


@pytest.mark.parametrize("cols", [96, 40, 33, 8, 2047])
def test_row_major_widths_that_are_no_multiple_of_a_tile(cols):
    """F = 96 / 40 / 33 / 8 / 2047 firings: partial last tiles, rows that start at any byte (the labels' 8-byte stores fall back to bytes)."""
    p = O.cfg_params("cfg2")
    fir = [u.synth_cloud(64, cols, 1, 7 + k) for k in range(3)]
    with u.Context(64 * cols, len(fir)) as ctx:
        nfir = fused_batch(ctx, fir, p)[2]   # (a sweep whose seam falls inside a tile is handed back in either layout)
        assert nfir >= 1 or cols == 33
    with u.Context(64 * cols, len(fir)) as ctx:
        rows_then_fused(ctx, [ring_major(c) for c in fir], p, want=nfir)
    # ragged: sweeps of different widths in one batch
    fir = [u.synth_cloud(64, c, 3, 11 + c) for c in (cols, 64, 2048 if cols > 2000 else 130)]
    with u.Context(max(len(s[0]) for s in fir), len(fir)) as ctx:
        nfir = fused_batch(ctx, fir, p, ragged=True)[2]
    with u.Context(max(len(s[0]) for s in fir), len(fir)) as ctx:
        rows_then_fused(ctx, [ring_major(c) for c in fir], p, want=nfir, ragged=True)


def test_a_row_major_sweep_that_is_not_clean_is_handed_back():
    """One point of a row lifted onto a neighbouring ring's angle: the rows' rule (every point on its row's table entry) fails in k_front,
    the table is walked the long way (k_table_repair) and the legacy kernels classify the sweep; its neighbours stay fused."""
    p = O.cfg_params("cfg2")
    x, y, z = (a.copy() for a in O.cfg_cloud("cfg2", 1))
    i, j = 20 + 64 * 700, 21 + 64 * 700            # firing 700: laser 20 gets laser 21's point
    x[i], y[i], z[i] = x[j], y[j], z[j]
    bad = ring_major((x, y, z))
    # a point that lies on NO ring of the table (between two rings): it would have become a leader of its own
    x2, y2, z2 = (a.copy() for a in O.cfg_cloud("cfg2", 2))
    k0, k1 = 30 + 64 * 900, 31 + 64 * 900
    z2[k0] = 0.5 * (z2[k0] + z2[k1] * np.hypot(x2[k0], y2[k0]) / max(np.hypot(x2[k1], y2[k1]), 1e-6))
    bad2 = ring_major((x2, y2, z2))
    scans = [ring_major(O.cfg_cloud("cfg2", 3)), bad, bad2, ring_major(O.cfg_cloud("sensor", 4))]
    with u.Context(N, len(scans)) as ctx:
        nf0, nf = rows_then_fused(ctx, scans, p)
        assert nf0 == 0 and 2 <= nf <= 3
        rows_then_fused(ctx, scans, p)   # (the legacy kernels now come as full grids)


@pytest.mark.parametrize("seed", range(24))
def test_row_major_organised_sweeps_with_holes(seed):
    """tests/fuzz_organised.py's sweeps (drop-outs, cut rings and firings, random regions of interest and parameters) stored row-major."""
    sw, p = case(7100000 + seed)
    scans = [ring_major(sw)]
    with u.Context(len(sw[0]), 1) as ctx:
        rows_then_fused(ctx, scans, p)


def test_row_major_ring_sorted_results_rerun_the_legacy_kernels():
    p = O.cfg_params("cfg2")
    scans = [ring_major(O.cfg_cloud("sensor", 1)), ring_major(O.cfg_cloud("narrow", 2))]
    with u.Context(N, 2) as ctx:
        X, Y, Z = (np.concatenate([s[k] for s in scans]) for k in range(3))
        dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
        dl = DevBuf(2 * N)
        ctx.set_params(p)
        ctx.set_front_mode(2)
        ctx.classify_batch_soa(dx, dy, dz, N, 2, dl, None)
        ctx.classify_batch_soa(dx, dy, dz, N, 2, dl, None)
        assert ctx.front_scans() == 2
        for k, (x, y, z) in enumerate(scans):
            lb, ib, st = O.run_b(x, y, z, p, debug=True)
            road, curb, prob = ctx.ordered_indices(N, scan=k)
            assert np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"]) and np.array_equal(prob, st["ring10_order"])
            assert np.array_equal(ctx.marker_points(scan=k), st["marker_pts"])
            assert np.array_equal(dl.to_numpy(np.uint8).reshape(2, N)[k], lb)
        assert ctx.front_scans() == 0


def test_row_major_after_the_other_speculations_were_switched_off():
    """A context whose look-ahead speculation has failed (rear-stored default-ROI sweeps in firing order) builds its tables the long way
    from then on; the rows' rule is independent of that: row-major sweeps still take the fused kernels from their second call."""
    pr = O.cfg_params("default_roi")
    rear = [rolled(O.cfg_cloud("default_roi", s), 1024) for s in (1, 2)]
    with u.Context(N, 2) as ctx:
        for call in range(3):
            labels, infos, nf = fused_batch(ctx, rear, pr)
            check_against_b(labels, infos, rear, pr)
            if nf == 2:
                break
        assert nf == 2 and call == 1   # no speculation any more: the tables are complete, both fused (and no row-major layout was sighted:
        #                                the first step of the walk that met a ring met dozens)
        rows = [ring_major(O.cfg_cloud("default_roi", s)) for s in (3, 4)]
        rows_then_fused(ctx, rows, pr, want=2)
        rows = [ring_major(c) for c in rear]
        labels, infos, nf = fused_batch(ctx, rows, pr)
        check_against_b(labels, infos, rows, pr)
        assert nf == 2


def test_small_row_major_batches_take_the_fused_kernels_in_the_default_mode():
    """Mode 1 (the default) starts the fused kernels at 192 sweeps per call -- for sweeps in firing order.  Row-major sweeps gain from them at any
    batch size: a small batch of them is sighted by its first call (general kernels), and from the second call on the context takes the fused
    kernels for small batches too; a context that only ever sees small batches in firing order stays with the general kernels."""
    p = O.cfg_params("cfg2")
    rows = [ring_major(O.cfg_cloud("cfg2", s)) for s in (1, 2)] + [ring_major(O.cfg_cloud("sensor", 3))]
    with u.Context(N, 3) as ctx:   # (mode 1 is the default)
        counts = []
        for call in range(4):
            labels, infos = run_batch(ctx, rows, p)
            check_against_b(labels, infos, rows, p)
            counts.append(ctx.front_scans())
        assert counts[0] == 0 and counts[-1] == 3, counts
    fir = [O.cfg_cloud("cfg2", s) for s in (1, 2, 3)]
    with u.Context(N, 3) as ctx:
        for call in range(3):
            labels, infos = run_batch(ctx, fir, p)
            check_against_b(labels, infos, fir, p)
            assert ctx.front_scans() == 0


def test_row_major_sweeps_on_the_callback_path():
    """One sweep per call (urf_classify_pc2: host message in, labels out).  Sweeps in firing order keep the general kernels there; a context whose
    sweeps come row-major sights the layout with its first sweep and takes the fused kernels from the second or third on -- inside the captured
    per-slot sequences, four sweeps in flight -- and the ring-sorted read-backs of such a sweep run it again through the general kernels."""
    p = O.cfg_params("cfg2")
    rows = [ring_major(O.cfg_cloud(name, s)) for name, s in (("cfg2", 1), ("sensor", 2), ("narrow", 3), ("cfg2", 4), ("sensor", 5))]
    ref = [O.run_b(*c, p, debug=True) for c in rows]
    with u.Context(N, 4, params=p) as ctx:
        fused = []
        for rep in range(3):
            for k, c in enumerate(rows):
                lab, info = ctx.classify_xyz(*c)
                assert np.array_equal(lab, ref[k][0]), (rep, k)
                assert info.n_road == ref[k][1]["n_road"] and info.n_curb == ref[k][1]["n_curb"] and info.n_ring_pts == ref[k][1]["n_ring_pts"]
                fused.append(ctx.front_scans())
        assert fused[0] == 0 and fused[-1] == 1 and sum(fused) >= 10, fused
        # four in flight
        recs = []
        for c in rows[:4]:
            r = np.zeros((N, 4), np.float32)
            r[:, 0], r[:, 1], r[:, 2] = c
            recs.append(r)
        for rep in range(2):
            tickets = [ctx.classify_pc2_async(r, N, 16, 0, 4, 8) for r in recs]
            for k, t in enumerate(tickets):
                lab = np.zeros(N, np.uint8)
                info = ctx.classify_pc2_wait(t, lab)
                assert np.array_equal(lab, ref[k][0]) and info.n_road == ref[k][1]["n_road"], (rep, k)
        # the published order of the last sweep: run again through the general kernels
        lab, info = ctx.classify_xyz(*rows[1])
        assert ctx.front_scans() == 1
        road, curb, prob = ctx.ordered_indices(N)
        st = ref[1][2]
        assert np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"]) and np.array_equal(prob, st["ring10_order"])
        assert np.array_equal(ctx.marker_points(), st["marker_pts"])
        lab, info = ctx.classify_xyz(*rows[2])                # ... and the context stays with them
        assert np.array_equal(lab, ref[2][0]) and ctx.front_scans() == 0
    fir = [O.cfg_cloud("cfg2", s) for s in (1, 2)]
    with u.Context(N, 4, params=p) as ctx:
        for rep in range(3):
            for c in fir:
                lab, info = ctx.classify_xyz(*c)
                assert np.array_equal(lab, O.run_b(*c, p)[0]) and ctx.front_scans() == 0
