"""The callback path (SURVEY.md 8b "wire in" / "callback"): one sweep arrives as PointCloud2 bytes in
host memory, the labels go back to host memory.  urf_classify_pc2_async / _wait keep two sweeps in
flight (pinned staging, H2D on a copy stream, the kernel sequence replayed from a captured graph);
urf_classify_pc2 is the same path, waited for at once.  Results must not depend on how a sweep was
submitted."""
import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u

pytestmark = pytest.mark.gpu
N = 64 * 2048


def records(x, y, z, step=32, ox=0, oy=4, oz=8):
    """pcl::PointXYZI wire layout: x y z pad intensity pad pad pad (32 bytes)."""
    buf = np.zeros((len(x), step), np.uint8)
    buf[:, ox:ox + 4] = x.view(np.uint8).reshape(-1, 4)
    buf[:, oy:oy + 4] = y.view(np.uint8).reshape(-1, 4)
    buf[:, oz:oz + 4] = z.view(np.uint8).reshape(-1, 4)
    return buf.reshape(-1)


@pytest.fixture(scope="module", params=[1, 2], ids=["one_scratch_row", "two_scratch_rows"])
def ctx(request):
    """max_batch 1: both slots share the scratch and the stream (only the copies overlap);
    max_batch 2: every slot has its own scratch row and compute stream (the kernels overlap too)."""
    c = u.Context(N, request.param, params=O.cfg_params("cfg2"))
    yield c
    c.close()


@pytest.fixture(scope="module")
def sweeps():
    p = O.cfg_params("cfg2")
    out = []
    for seed in range(40, 46):
        x, y, z = O.cfg_cloud("cfg2" if seed % 2 else "narrow", seed)
        lb, ib, _ = O.run_b(x, y, z, p)
        out.append((records(x, y, z), lb, ib))
    return out


def test_two_in_flight_equal_the_oracle(ctx, sweeps):
    ctx.set_params(O.cfg_params("cfg2"))
    got = [None] * len(sweeps)
    tickets = []
    for k, (rec, _, _) in enumerate(sweeps):
        if len(tickets) == 2:   # both slots busy: a third submission is refused, nothing is lost
            with pytest.raises(u.UrfError) as e:
                ctx.classify_pc2_async(rec, N, 32, 0, 4, 8)
            assert e.value.code == -7
            j, t = tickets.pop(0)
            lab = np.empty(N, np.uint8)
            info = ctx.classify_pc2_wait(t, lab)
            got[j] = (lab, info)
        tickets.append((k, ctx.classify_pc2_async(rec, N, 32, 0, 4, 8)))
    for j, t in tickets:
        lab = np.empty(N, np.uint8)
        got[j] = (lab, ctx.classify_pc2_wait(t, lab))
    for (rec, lb, ib), (lab, info) in zip(sweeps, got):
        assert np.array_equal(lab, lb)
        assert (info.n_road, info.n_curb, info.n_roi, info.n_rings) == (ib["n_road"], ib["n_curb"], ib["n_roi"], ib["n_rings"])


def test_sync_entry_point_is_the_same_path(ctx, sweeps):
    ctx.set_params(O.cfg_params("cfg2"))
    for rec, lb, ib in sweeps[:3]:
        lab, info = ctx.classify_pc2(rec, N, 32, 0, 4, 8)
        assert np.array_equal(lab, lb) and info.n_road == ib["n_road"]


def test_producer_writes_into_the_pinned_buffer(ctx, sweeps):
    """zero-copy submission: the message is produced in the slot's pinned buffer; the result is read
    in place from the pinned result buffer."""
    ctx.set_params(O.cfg_params("cfg2"))
    rec, lb, _ = sweeps[1]
    seen = set()
    for rep in range(3):   # both slots get used
        buf = ctx.pinned_input(len(rec))
        seen.add(buf.ctypes.data)
        buf[:] = rec
        t = ctx.classify_pc2_async(buf.ctypes.data, N, 32, 0, 4, 8)
        info = ctx.classify_pc2_wait(t)
        assert info.status == 0 and np.array_equal(ctx.result_labels(t, N), lb)
    assert len(seen) == 2


def test_graph_replay_equals_kernel_by_kernel_launches(ctx, sweeps):
    ctx.set_params(O.cfg_params("cfg2"))
    rec, lb, _ = sweeps[2]
    a, _ = ctx.classify_pc2(rec, N, 32, 0, 4, 8)
    ctx.set_debug_flags(8)   # no graph
    try:
        b, _ = ctx.classify_pc2(rec, N, 32, 0, 4, 8)
    finally:
        ctx.set_debug_flags(0)
    c, _ = ctx.classify_pc2(rec, N, 32, 0, 4, 8)
    assert np.array_equal(a, lb) and np.array_equal(b, lb) and np.array_equal(c, lb)


def test_a_captured_sequence_follows_parameter_and_shape_changes(ctx):
    """set_params between callbacks (paramsCallback, main.cpp:4-34), another message layout and a
    shorter sweep: the captured sequence is rebuilt, the results follow."""
    p = O.cfg_params("cfg2")
    x, y, z = O.cfg_cloud("cfg2", 77)
    for tweak in ({}, {"curbHeight": 0.1}, {"blind_spots": 0}, {"star_shaped_method": 0}, {}):
        q = p.copy()
        for k, v in tweak.items():
            setattr(q, k, v)
        ctx.set_params(q)
        lb, _, _ = O.run_b(x, y, z, q)
        lab, _ = ctx.classify_pc2(records(x, y, z), len(x), 32, 0, 4, 8)
        assert np.array_equal(lab, lb), tweak
    ctx.set_params(p)
    lb, _, _ = O.run_b(x, y, z, p)
    lab, _ = ctx.classify_pc2(records(x, y, z, step=48, ox=20, oy=4, oz=36), len(x), 48, 20, 4, 36)
    assert np.array_equal(lab, lb)
    m = 50000
    lb, _, _ = O.run_b(x[:m], y[:m], z[:m], p)
    lab, _ = ctx.classify_pc2(records(x[:m], y[:m], z[:m]), m, 32, 0, 4, 8)
    assert np.array_equal(lab, lb)


def test_malformed_layouts_are_refused_before_any_copy(ctx):
    rec = np.zeros(32 * 100, np.uint8)
    for step, ox, oy, oz in ((32, 0xFFFFFFFE, 4, 8), (32, 0, 30, 8), (3, 0, 0, 0), (32, 0, 4, 29)):
        with pytest.raises(u.UrfError) as e:
            ctx.classify_pc2(rec, 100, step, ox, oy, oz)
        assert e.value.code == -1
