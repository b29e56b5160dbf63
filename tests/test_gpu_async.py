"""The callback path (SURVEY.md 8b "wire in" / "callback"): one sweep arrives as PointCloud2 bytes in
host memory, the labels go back to host memory.  urf_classify_pc2_async / _wait keep up to four sweeps in
flight (pinned staging, H2D and the kernel sequence -- replayed from a captured graph -- on the slot's stream);
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


IN_FLIGHT = 4   # URF_MAX_IN_FLIGHT


@pytest.fixture(scope="module", params=[1, 2, 4], ids=["one_scratch_row", "two_scratch_rows", "four_scratch_rows"])
def ctx(request):
    """max_batch 1: all slots share the scratch and the stream (only the copies overlap);
    max_batch 2 / 4: the slots are spread over that many scratch rows, each with its own compute stream
    (the kernels overlap too)."""
    c = u.Context(N, request.param, params=O.cfg_params("cfg2"))
    yield c
    c.close()


@pytest.fixture(scope="module")
def hctx():
    """A context in liburf_hip_test.so (the product's sources + include/urf_test_hooks.h): debug flags, the library-side loop."""
    c = u.Context(N, 4, params=O.cfg_params("cfg2"), hooks=True)
    yield c
    c.close()


@pytest.fixture(scope="module")
def sweeps():
    p = O.cfg_params("cfg2")
    out = []
    for seed in range(40, 50):
        x, y, z = O.cfg_cloud("cfg2" if seed % 2 else "narrow", seed)
        lb, ib, _ = O.run_b(x, y, z, p)
        out.append((records(x, y, z), lb, ib))
    return out


def test_sweeps_in_flight_equal_the_oracle(ctx, sweeps):
    ctx.set_params(O.cfg_params("cfg2"))
    got = [None] * len(sweeps)
    tickets = []
    for k, (rec, _, _) in enumerate(sweeps):
        if len(tickets) == IN_FLIGHT:   # every slot busy: one more submission is refused, nothing is lost
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
    for rep in range(IN_FLIGHT + 1):   # every slot gets used
        buf = ctx.pinned_input(len(rec))
        seen.add(buf.ctypes.data)
        buf[:] = rec
        t = ctx.classify_pc2_async(buf.ctypes.data, N, 32, 0, 4, 8)
        info = ctx.classify_pc2_wait(t)
        assert info.status == 0 and np.array_equal(ctx.result_labels(t, N), lb)
    assert len(seen) == IN_FLIGHT


def test_tickets_and_pinned_pointers_are_validated(ctx, sweeps):
    """urf_result_labels refuses a ticket that was never issued, is still in flight or has been overtaken;
    a message that lies inside the slot's pinned buffer must be exactly that buffer."""
    ctx.set_params(O.cfg_params("cfg2"))
    rec, lb, _ = sweeps[0]
    t = ctx.classify_pc2_async(rec, N, 32, 0, 4, 8)
    with pytest.raises(u.UrfError) as e:
        ctx.result_labels(t, N)
    assert e.value.code == -7            # in flight
    with pytest.raises(u.UrfError) as e:
        ctx.result_labels(t + 1000, N)
    assert e.value.code == -1            # never issued
    ctx.classify_pc2_wait(t)
    assert np.array_equal(ctx.result_labels(t, N), lb)
    for _ in range(IN_FLIGHT):           # the slot is used again: the old ticket is stale
        ctx.classify_pc2(rec, N, 32, 0, 4, 8)
    with pytest.raises(u.UrfError) as e:
        ctx.result_labels(t, N)
    assert e.value.code == -1
    buf = ctx.pinned_input(len(rec))
    with pytest.raises(u.UrfError) as e:   # starts inside the pinned buffer, but is not the buffer
        ctx.classify_pc2_async(buf.ctypes.data + 32, N - 1, 32, 0, 4, 8)
    assert e.value.code == -1
    small = u.Context(N, 1, params=O.cfg_params("cfg2"))
    try:
        b2 = small.pinned_input(32 * 1000)
        with pytest.raises(u.UrfError) as e:   # longer than the size asked for
            small.classify_pc2_async(b2.ctypes.data, N, 32, 0, 4, 8)
        assert e.value.code == -1
    finally:
        small.close()


def test_batch_calls_and_readbacks_are_ordered_behind_sweeps_in_flight(sweeps):
    """a batch call overwrites the scratch rows (and the SoA staging) that sweeps of the callback path
    still in flight on their own streams work on; the read-back entry points read them.  All of them must
    wait."""
    p = O.cfg_params("cfg2")
    import hipmem as H
    with u.Context(N, 4, params=p) as c:
        xs = [O.cfg_cloud("cfg2", 60 + k) for k in range(4)]
        X = np.concatenate([v[0] for v in xs]); Y = np.concatenate([v[1] for v in xs]); Z = np.concatenate([v[2] for v in xs])
        want = [O.run_b(*v, p)[0] for v in xs]
        dx, dy, dz = H.DevBuf.from_numpy(X), H.DevBuf.from_numpy(Y), H.DevBuf.from_numpy(Z)
        dl = H.DevBuf(4 * N)
        for rep in range(3):
            ts = [c.classify_pc2_async(sweeps[k][0], N, 32, 0, 4, 8) for k in range(IN_FLIGHT)]
            c.classify_batch_soa(dx, dy, dz, N, 4, dl, None)   # rows 0..3 while four sweeps are in flight
            got = []
            for t in ts:
                lab = np.empty(N, np.uint8)
                c.classify_pc2_wait(t, lab)
                got.append(lab)
            c.synchronize()
            L = dl.to_numpy(np.uint8)
            for k in range(4):
                assert np.array_equal(got[k], sweeps[k][1]), (rep, k)
                assert np.array_equal(L[k * N:(k + 1) * N], want[k]), (rep, k)
        # read-back right after a submission on a row with its own stream
        t0 = c.classify_pc2_async(sweeps[0][0], N, 32, 0, 4, 8)
        t1 = c.classify_pc2_async(sweeps[1][0], N, 32, 0, 4, 8)
        c.classify_pc2_wait(t0)
        c.classify_pc2_wait(t1)
        det = c.read_stage(u.STAGE_DETECT, N)
        assert det.shape == (N,)
        road, curb, r10 = c.ordered_indices(N)
        assert set(curb.tolist()) == set(np.nonzero((sweeps[1][1] & 3) == 2)[0].tolist())
        for b in (dx, dy, dz, dl):
            b.free()


def test_graph_replay_equals_kernel_by_kernel_launches(hctx, sweeps):
    ctx = hctx
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


@pytest.mark.parametrize("pinned", [False, True], ids=["staged", "pinned_producer"])
def test_native_submit_collect_loop(hctx, sweeps, pinned):
    """urf_bench_callback_stream (include/urf_test_hooks.h): the loop a C / C++ client runs (submit until IN_FLIGHT sweeps are in flight, collect
    the oldest), inside the library.  The labels it hands back are those of the last message."""
    ctx = hctx
    ctx.set_params(O.cfg_params("cfg2"))
    msgs = [rec for rec, _, _ in sweeps[:5]]
    for in_flight in (1, 2, IN_FLIGHT):
        n_sweeps = 11
        sec, lab = ctx.bench_callback_stream(msgs, N, 32, 0, 4, 8, n_sweeps, in_flight, producer_pinned=pinned)
        assert sec > 0
        if pinned:   # each slot's pinned buffer was filled once, with message k for the k-th submission (k < IN_FLIGHT)
            want = sweeps[(n_sweeps - 1) % IN_FLIGHT][1]
        else:
            want = sweeps[(n_sweeps - 1) % len(msgs)][1]
        assert np.array_equal(lab, want), (in_flight, pinned)
    with pytest.raises(u.UrfError) as e:
        ctx.bench_callback_stream(msgs, N, 32, 0, 4, 8, 4, IN_FLIGHT + 1)
    assert e.value.code == -1


def _in_flight(ctx, msgs, n):
    """Submit all messages, at most IN_FLIGHT at a time; labels and summaries in submission order."""
    got, tickets = [None] * len(msgs), []
    for k, rec in enumerate(msgs):
        if len(tickets) == IN_FLIGHT:
            j, t = tickets.pop(0)
            lab = np.empty(n[j], np.uint8)
            got[j] = (lab, ctx.classify_pc2_wait(t, lab))
        tickets.append((k, ctx.classify_pc2_async(rec, n[k], 32, 0, 4, 8)))
    for j, t in tickets:
        lab = np.empty(n[j], np.uint8)
        got[j] = (lab, ctx.classify_pc2_wait(t, lab))
    return got


@pytest.mark.parametrize("rows", [1, 4])
def test_a_sweep_the_short_sequence_cannot_handle_is_run_again(rows):
    """The callback path launches without the repair kernels of the speculative ring table and without the kernels
    for oversized star sectors; k_index voids a sweep that needed them (internal status) and urf_classify_pc2_wait runs
    it again with the full sequence -- as every later sweep.  Here: (i) a sweep with 6 000 extra points in two sectors
    (mid and big work lists) among ordinary ones, several in flight when the first one comes back void; (ii) in a fresh
    context a default-ROI sweep stored from the rear (no region-of-interest point among its first 8192: the speculative
    table is empty).  Nobody ever sees the internal status, all labels equal oracle B."""
    from test_gpu_parity import crowded_cloud
    p = O.cfg_params("cfg2")
    plain = [O.cfg_cloud("cfg2", 60 + k) for k in range(3)]
    xc, yc, zc = crowded_cloud(6000, seed=61)
    crowd = tuple(np.concatenate([a, b]) for a, b in zip(O.cfg_cloud("cfg2", 62), (xc, yc, zc)))
    order = [plain[0], crowd, plain[1], crowd, plain[2], plain[0]]
    with u.Context(N + 6000, rows, params=p) as ctx:
        got = _in_flight(ctx, [records(*c) for c in order], [len(c[0]) for c in order])
        for c, (lab, info) in zip(order, got):
            lb, ib, _ = O.run_b(*c, p)
            assert info.status == 0 and np.array_equal(lab, lb)
            assert (info.n_road, info.n_curb, info.n_roi) == (ib["n_road"], ib["n_curb"], ib["n_roi"])
        # both crowded sweeps had been launched with the short sequence when the first one came back void; the ring
        # table is still speculative
        assert ctx.callback_path_state() == (2, 1 | 2 | 8)   # (8: the ring table also stops at the previous sweep's ring count)
    q = O.cfg_params("default_roi")
    front = O.cfg_cloud("default_roi", 63)
    rear = tuple(np.roll(a.reshape(-1, 64), 1024, axis=0).reshape(-1).copy() for a in O.cfg_cloud("default_roi", 64))
    order = [front, rear, front, rear, front]
    with u.Context(N, rows, params=q) as ctx:
        for rep in range(2):
            got = _in_flight(ctx, [records(*c) for c in order], [N] * len(order))
            for c, (lab, info) in zip(order, got):
                lb, ib, _ = O.run_b(*c, q)
                assert info.status == 0 and np.array_equal(lab, lb)
                assert (info.n_road, info.n_curb, info.n_rings) == (ib["n_road"], ib["n_curb"], ib["n_rings"])
        n_rerun, seq = ctx.callback_path_state()
        # four rows: the second rear sweep was in flight when the first came back void, and is run again as well
        assert seq == 0 and 1 <= n_rerun <= 2


def test_sweeps_of_oversized_sectors_through_the_callback_path():
    """128 x 4096 sweeps (every star sector on the mid work list): the first one comes back void and is run again,
    the others take the full sequence at once."""
    p = O.cfg_params("cfg5")
    clouds = [O.cfg_cloud("cfg5", 70 + k) for k in range(3)]
    n = len(clouds[0][0])
    with u.Context(n, 2, params=p) as ctx:
        lab, info = ctx.classify_pc2(records(*clouds[0]), n, 32, 0, 4, 8)
        lb, _, _ = O.run_b(*clouds[0], p)
        assert info.status == 0 and np.array_equal(lab, lb) and ctx.callback_path_state() == (1, 1 | 2 | 8)
        got = _in_flight(ctx, [records(*c) for c in clouds], [n] * 3)
        for c, (lab, info) in zip(clouds, got):
            lb, ib, _ = O.run_b(*c, p)
            assert info.status == 0 and np.array_equal(lab, lb)
        assert ctx.callback_path_state() == (1, 1 | 2 | 8)


@pytest.mark.parametrize("step,ox,oy,oz", [(32, 0, 4, 8), (12, 0, 4, 8), (16, 4, 8, 12), (20, 4, 8, 12), (48, 20, 4, 36), (16, 8, 4, 0)])
def test_staged_messages_of_any_layout(ctx, step, ox, oy, oz):
    """A staged message crosses PCIe as three planes (the host gathers x / y / z: four records at a time where they
    lie side by side with a fourth word behind them, one by one otherwise); a producer that fills the pinned buffer
    sends the records and the device gathers.  Same labels either way, for sweep lengths that are no multiple of 4."""
    p = O.cfg_params("cfg2")
    ctx.set_params(p)
    x, y, z = O.cfg_cloud("cfg2", 81)
    for m in (N, 50001, 50002, 50003):
        lb, _, _ = O.run_b(x[:m], y[:m], z[:m], p)
        rec = records(x[:m], y[:m], z[:m], step=step, ox=ox, oy=oy, oz=oz)
        lab, info = ctx.classify_pc2(rec, m, step, ox, oy, oz)
        assert info.status == 0 and np.array_equal(lab, lb), m
        pin = ctx.pinned_input(rec.size)
        pin[:] = rec
        lab2 = np.empty(m, np.uint8)
        ctx.classify_pc2_wait(ctx.classify_pc2_async(pin, m, step, ox, oy, oz), lab2)
        assert np.array_equal(lab2, lb), m
