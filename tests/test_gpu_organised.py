"""Organised sweeps with holes (tests/fuzz_organised.py): the HIP path against oracle B -- labels, detector stage, summary, and for
a few cases the published order and the marker points."""
import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u
from fuzz_organised import case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = u.Context(64 * 2048, 1)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(48))
def test_organised_sweeps_with_holes(ctx, seed):
    (x, y, z), p = case(7_100_000 + seed)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx.set_params(p)
    lg, ig = ctx.classify_xyz(x, y, z)
    assert np.array_equal(lg, lb), "%d labels differ" % int((lg != lb).sum())
    assert all(getattr(ig, k) == ib[k] for k in ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10"))
    if ib["status"] == 0:
        assert np.array_equal(ctx.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
        if seed % 6 == 0:
            road, curb, prob = ctx.ordered_indices(len(x))
            assert np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"]) and np.array_equal(prob, st["ring10_order"])
            mg = ctx.marker_points()
            assert mg.shape == st["marker_pts"].shape and np.array_equal(mg, st["marker_pts"])


def test_organised_sweeps_with_holes_in_a_batch():
    """... and twelve of them (2048 columns each) in one batch call."""
    from hipmem import DevBuf
    cases = []
    s = 7_200_000
    while len(cases) < 12:
        (x, y, z), p = case(s)
        s += 1
        if len(x) == 64 * 2048:
            cases.append((x, y, z))
    p = O.cfg_params("cfg2")
    X, Y, Z = (np.concatenate([c[k] for c in cases]) for k in range(3))
    dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
    dl = DevBuf(len(X))
    with u.Context(64 * 2048, 12, params=p) as c:
        c.classify_batch_soa(dx, dy, dz, 64 * 2048, 12, dl, None)
        L = dl.to_numpy(np.uint8).reshape(12, -1)
    for k, (x, y, z) in enumerate(cases):
        lb, _, _ = O.run_b(x, y, z, p)
        assert np.array_equal(L[k], lb), k
