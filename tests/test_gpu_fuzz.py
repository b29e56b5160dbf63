"""Differential fuzzing on the GPU: the HIP path against oracle B on random unorganised clouds
and random parameters (tests/fuzz.py), single scans and one ragged batch of all of them."""
import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u
from fuzz import case
from hipmem import DevBuf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = u.Context(32768, 1)
    yield c
    c.close()


def check_case(ctx, seed, capture):
    (x, y, z), p = case(2000 + seed)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx.set_params(p)
    if not capture:   # the production path proper (records nothing per input point)
        lg, ig = ctx.classify_xyz(x, y, z)
        assert np.array_equal(lg, lb), "seed %d: %d labels differ" % (seed, int((lg != lb).sum()))
    ctx.enable_stage_capture(1 if capture else 2)   # 2: the production decisions, ring / sector recorded
    try:
        lg, ig = ctx.classify_xyz(x, y, z)
        n = len(x)
        if ib["status"] == 0:
            assert np.array_equal(ctx.read_stage(u.STAGE_RING, n), st["ring"])
            if p.star_shaped_method:
                assert np.array_equal(ctx.read_stage(u.STAGE_SECTOR, n), st["sector"])
            assert np.array_equal(ctx.read_stage(u.STAGE_DETECT, n), st["detect"])
            assert np.array_equal(ctx.read_stage(u.STAGE_BEAM_STOP, n), st["beam_stop"])
    finally:
        ctx.enable_stage_capture(0)
    assert np.array_equal(lg, lb), "seed %d: %d labels differ" % (seed, int((lg != lb).sum()))
    assert all(getattr(ig, k) == ib[k] for k in ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10"))


@pytest.mark.parametrize("seed", range(200))
def test_hip_equals_oracle_on_random_input(ctx, seed):
    """Production configuration: decisions on float approximations wherever they clear the margins."""
    check_case(ctx, seed, capture=False)


@pytest.mark.parametrize("seed", range(0, 200, 4))
def test_hip_equals_oracle_on_random_input_exact_mode(ctx, seed):
    """Stage capture on: every point takes the reference's exact arithmetic."""
    check_case(ctx, seed, capture=True)


def test_ragged_batch_of_random_scans():
    p = O.cfg_params("cfg2")
    p.interval = 0.5
    scans = [case(3000 + s)[0] for s in range(24)]
    lens = [len(s[0]) for s in scans]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    X, Y, Z = (np.concatenate([s[k] for s in scans]) for k in range(3))
    dx, dy, dz, do = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z), DevBuf.from_numpy(offs)
    dl = DevBuf(len(X))
    with u.Context(max(lens), len(scans), params=p) as ctx:
        ctx.classify_batch_soa_ragged(dx, dy, dz, do, max(lens), len(scans), dl, None)
        L = dl.to_numpy(np.uint8)
    for k, (x, y, z) in enumerate(scans):
        lb, _, _ = O.run_b(x, y, z, p)
        assert np.array_equal(L[offs[k]:offs[k + 1]], lb), "scan %d" % k
