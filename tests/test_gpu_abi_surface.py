"""Entry points of include/urf.h that no other test reached (VERDICT r3, weak #2 and hygiene): urf_classify_batch_pc2,
urf_set_stream with a caller-owned stream, several contexts in one process, and what the callback path does when more
sweeps are in flight than the context has scratch rows."""
import ctypes as C

import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u
from hipmem import DevBuf, hip
from test_gpu_async import records
from test_gpu_parity import check_against_b

pytestmark = pytest.mark.gpu
N = 64 * 2048


@pytest.mark.parametrize("step,ox,oy,oz", [(23, 3, 11, 17), (48, 20, 4, 36), (16, 0, 4, 8)],
                         ids=["step23_unaligned", "step48_permuted", "step16_xyzi"])
def test_classify_batch_pc2(step, ox, oy, oz):
    """Three device-resident PointCloud2-layout scans back to back (records -> SoA on the device, then the batch
    pipeline): labels and summaries equal oracle B scan by scan."""
    p = O.cfg_params("cfg2")
    scans = [O.cfg_cloud("cfg2", 71), O.cfg_cloud("narrow", 72), O.cfg_cloud("cfg2", 73)]
    raw = np.concatenate([records(x, y, z, step=step, ox=ox, oy=oy, oz=oz) for x, y, z in scans])
    d_raw = DevBuf.from_numpy(raw)
    dl = DevBuf(N * len(scans))
    dl.fill(0xEE)
    di = DevBuf(32 * len(scans))
    with u.Context(N, len(scans), params=p) as ctx:
        ctx.classify_batch_pc2(d_raw, N, len(scans), step, ox, oy, oz, dl, di)
        ctx.synchronize()
        L = dl.to_numpy(np.uint8).reshape(len(scans), N)
        infos = di.to_numpy(np.uint32).reshape(len(scans), 8)
        check_against_b(list(L), infos, scans, p)
        # the call is "the last call": its published order can be asked for (scan 1)
        lb, ib, st = O.run_b(*scans[1], p, debug=True)
        road, curb, prob = ctx.ordered_indices(N, scan=1)
        assert np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"])
        # capacity and layout errors
        with pytest.raises(u.UrfError) as e:
            ctx.classify_batch_pc2(d_raw, N, len(scans) + 1, step, ox, oy, oz, dl, di)
        assert e.value.code == -4
        with pytest.raises(u.UrfError) as e:
            ctx.classify_batch_pc2(d_raw, N, 1, step, step - 2, oy, oz, dl, di)
        assert e.value.code == -1
    for b in (d_raw, dl, di):
        b.free()


def _stream_create():
    h = hip()
    h.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    h.hipStreamSynchronize.argtypes = [C.c_void_p]
    h.hipStreamDestroy.argtypes = [C.c_void_p]
    h.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    h.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    st = C.c_void_p()
    assert h.hipStreamCreateWithFlags(C.byref(st), 1) == 0   # hipStreamNonBlocking
    return st


def test_set_stream_orders_the_batch_behind_the_callers_work():
    """urf_set_stream: the batch call runs on a caller-owned stream, in order with the caller's own operations on it.
    The inputs reach the device by asynchronous copies queued on that stream immediately in front of the call, the
    labels leave by one queued behind it -- nothing but the stream's own order makes that right -- while four sweeps of
    the callback path are in flight on the context's other streams.  NULL restores the context's stream."""
    p = O.cfg_params("cfg2")
    h = hip()
    st = _stream_create()
    scans = [O.cfg_cloud("cfg2", 81), O.cfg_cloud("narrow", 82)]
    X, Y, Z = (np.concatenate([s[k] for s in scans]) for k in range(3))
    dx, dy, dz = DevBuf(X.nbytes), DevBuf(Y.nbytes), DevBuf(Z.nbytes)
    dl = DevBuf(2 * N)
    out = np.zeros(2 * N, np.uint8)
    sweeps = [O.cfg_cloud("cfg2", 90 + k) for k in range(4)]
    recs = [records(*s) for s in sweeps]
    with u.Context(N, 4, params=p) as ctx:
        for rounds in range(3):
            for b in (dx, dy, dz):
                assert h.hipMemsetAsync(b.ptr, 0, b.nbytes, st) == 0   # stale inputs unless the copies below come first
            assert h.hipMemsetAsync(dl.ptr, 0xEE, dl.nbytes, st) == 0
            tickets = [ctx.classify_pc2_async(r, N, 32, 0, 4, 8) for r in recs]   # four sweeps in flight on the row streams
            ctx.set_stream(st.value)
            for b, a in ((dx, X), (dy, Y), (dz, Z)):
                assert h.hipMemcpyAsync(b.ptr, a.ctypes.data, a.nbytes, 1, st) == 0
            ctx.classify_batch_soa(dx, dy, dz, N, 2, dl, None)
            assert h.hipMemcpyAsync(out.ctypes.data, dl.ptr, out.nbytes, 2, st) == 0
            assert h.hipStreamSynchronize(st) == 0
            for k, s in enumerate(scans):
                lb, _, _ = O.run_b(*s, p)
                assert np.array_equal(out[k * N:(k + 1) * N], lb), (rounds, k)
            ctx.set_stream(None)
            for k, t in enumerate(tickets):
                lab = np.empty(N, np.uint8)
                ctx.classify_pc2_wait(t, lab)
                lb, _, _ = O.run_b(*sweeps[k], p)
                assert np.array_equal(lab, lb), (rounds, k)
        # back on the context's own stream
        dl.fill(0xEE)
        ctx.classify_batch_soa(dx, dy, dz, N, 2, dl, None)
        ctx.synchronize()
        lb, _, _ = O.run_b(*scans[1], p)
        assert np.array_equal(dl.to_numpy(np.uint8)[N:], lb)
    assert h.hipStreamDestroy(st) == 0
    for b in (dx, dy, dz, dl):
        b.free()


def test_two_contexts_in_one_process_are_independent():
    """include/urf.h: "any number of contexts may coexist".  Two contexts on device 0 with different parameters,
    interleaved calls (single sweeps, a batch, read-backs): neither sees the other's parameters or scratch."""
    pa = O.cfg_params("cfg2")
    pb = O.cfg_params("default_roi")
    pb.curbPoints = 9
    a = O.cfg_cloud("cfg2", 101)
    b = O.cfg_cloud("narrow", 102)
    with u.Context(N, 2, params=pa) as ca, u.Context(N, 1, params=pb) as cb:
        ta = ca.classify_pc2_async(records(*a), N, 32, 0, 4, 8)
        tb = cb.classify_pc2_async(records(*b), N, 32, 0, 4, 8)
        la, lbb = np.empty(N, np.uint8), np.empty(N, np.uint8)
        ia = ca.classify_pc2_wait(ta, la)
        ib = cb.classify_pc2_wait(tb, lbb)
        wa, wia, sta = O.run_b(*a, pa, debug=True)
        wb, wib, stb = O.run_b(*b, pb, debug=True)
        assert np.array_equal(la, wa) and np.array_equal(lbb, wb)
        assert ia.n_road == wia["n_road"] and ib.n_road == wib["n_road"] and ib.n_roi == wib["n_roi"]
        # read-backs of context a while context b classifies something else
        cb.classify_xyz(*a)
        road, curb, _ = ca.ordered_indices(N)
        assert np.array_equal(road, sta["road_order"]) and np.array_equal(curb, sta["curb_order"])
        assert np.array_equal(ca.read_stage(u.STAGE_MAXDIST, N)[:wia["n_rings"]], sta["max_dist"][:wia["n_rings"]])
        lab2, _ = cb.classify_xyz(*b)
        assert np.array_equal(lab2, wb)
        assert ca.get_params().curbPoints == 5 and cb.get_params().curbPoints == 9


@pytest.mark.parametrize("rows", [1, 2])
def test_more_sweeps_in_flight_than_scratch_rows(rows):
    """max_batch 1 / 2 and four sweeps in flight: slots share scratch rows.  Labels and summaries of every sweep are
    right (each slot has its own result buffers); the read-backs that look at a sweep's ROW refuse (URF_ERR_BUSY) once a
    later sweep has been submitted on it instead of mixing two sweeps -- and work again for the row's latest sweep."""
    p = O.cfg_params("cfg2")
    sweeps = [O.cfg_cloud("cfg2" if k % 2 else "narrow", 110 + k) for k in range(4)]
    want = [O.run_b(*s, p, debug=True) for s in sweeps]
    with u.Context(N, rows, params=p) as ctx:
        tickets = [ctx.classify_pc2_async(records(*s), N, 32, 0, 4, 8) for s in sweeps]
        for k, t in enumerate(tickets):
            lab = np.empty(N, np.uint8)
            info = ctx.classify_pc2_wait(t, lab)
            assert np.array_equal(lab, want[k][0]), k
            assert info.n_road == want[k][1]["n_road"] and info.n_curb == want[k][1]["n_curb"]
            latest_on_row = k + rows >= len(sweeps)
            if latest_on_row:
                road, curb, _ = ctx.ordered_indices(N)
                assert np.array_equal(road, want[k][2]["road_order"]) and np.array_equal(curb, want[k][2]["curb_order"]), k
                assert np.array_equal(ctx.marker_points(), want[k][2]["marker_pts"])
            else:
                for call in (lambda: ctx.ordered_indices(N), lambda: ctx.marker_points(),
                             lambda: ctx.read_stage(u.STAGE_QUADRANTS, N)):
                    with pytest.raises(u.UrfError) as e:
                        call()
                    assert e.value.code == -7, k
        # a fresh sweep alone on its row reads back fine again
        lab, _ = ctx.classify_xyz(*sweeps[0])
        assert np.array_equal(lab, want[0][0])
        road, _, _ = ctx.ordered_indices(N)
        assert np.array_equal(road, want[0][2]["road_order"])


def test_rerun_of_a_voided_sweep_counts_as_the_rows_latest_submission():
    """One scratch row, two sweeps in flight, the FIRST one needs a kernel the short launch sequence leaves out (a ring
    point on the sensor's axis: k_nan_rings): its rerun inside urf_classify_pc2_wait() is queued behind the second sweep
    and overwrites the row.  The read-backs of the second sweep must then refuse (URF_ERR_BUSY) instead of returning the
    first sweep's intermediates; the first sweep's own read-backs are served (the row holds its rerun)."""
    p = O.cfg_params("cfg2")
    first = list(O.cfg_cloud("cfg2", 120))
    for a, v in zip(first, (0.0, 0.0, -1.8)):
        a[5000] = v                                   # x == y == 0 on a ring: NaN azimuth
    second = O.cfg_cloud("narrow", 121)
    wa, wb = O.run_b(*first, p, debug=True), O.run_b(*second, p, debug=True)
    with u.Context(N, 1, params=p) as ctx:
        ta = ctx.classify_pc2_async(records(*first), N, 32, 0, 4, 8)
        tb = ctx.classify_pc2_async(records(*second), N, 32, 0, 4, 8)
        lab = np.empty(N, np.uint8)
        ctx.classify_pc2_wait(ta, lab)
        assert np.array_equal(lab, wa[0]) and ctx.callback_path_state()[0] >= 1
        road, curb, _ = ctx.ordered_indices(N)        # the rerun was the row's last submission
        assert np.array_equal(road, wa[2]["road_order"]) and np.array_equal(curb, wa[2]["curb_order"])
        ctx.classify_pc2_wait(tb, lab)
        assert np.array_equal(lab, wb[0])             # labels and summary live in the slot's own buffers
        for call in (lambda: ctx.ordered_indices(N), lambda: ctx.marker_points(), lambda: ctx.read_stage(u.STAGE_QUADRANTS, N)):
            with pytest.raises(u.UrfError) as e:
                call()
            assert e.value.code == -7


def test_classify_batch_pc2_row_major_and_firing_order_take_the_fused_kernels():
    """PointCloud2 records of organised sweeps through urf_classify_batch_pc2 in front mode 2: the records -> SoA kernel feeds the fused
    front end like any caller's arrays; the row-major scan (an organised cloud, height = 64) is sighted by the first call and fused from
    the second; the ring-sorted read-backs afterwards run the call again from the context's own SoA copy."""
    p = O.cfg_params("cfg2")
    fir = O.cfg_cloud("sensor", 81)
    rows = tuple(np.ascontiguousarray(a.reshape(-1, 64).T.reshape(-1)) for a in O.cfg_cloud("cfg2", 82))
    scans = [fir, rows, O.cfg_cloud("narrow", 83)]
    step, ox, oy, oz = 32, 0, 4, 8
    raw = np.concatenate([records(x, y, z, step=step, ox=ox, oy=oy, oz=oz) for x, y, z in scans])
    d_raw = DevBuf.from_numpy(raw)
    dl = DevBuf(N * len(scans))
    di = DevBuf(32 * len(scans))
    with u.Context(N, len(scans), params=p) as ctx:
        ctx.set_front_mode(2)
        fused = []
        for call in range(3):
            dl.fill(0xEE)
            ctx.classify_batch_pc2(d_raw, N, len(scans), step, ox, oy, oz, dl, di)
            ctx.synchronize()
            fused.append(ctx.front_scans())
            L = dl.to_numpy(np.uint8).reshape(len(scans), N)
            infos = di.to_numpy(np.uint32).reshape(len(scans), 8)
            check_against_b(list(L), infos, scans, p)
        assert fused == [2, 3, 3], fused
        lb, ib, st = O.run_b(*scans[1], p, debug=True)
        road, curb, prob = ctx.ordered_indices(N, scan=1)
        assert np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"])
    for b in (d_raw, dl, di):
        b.free()
