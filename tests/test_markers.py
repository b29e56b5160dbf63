"""road_marker: marker points on the GPU, line strips on the host.
  CPU: oracle B (marker points + strips, C) against the reference binary's published MarkerArray over
       sequences of sweeps (ghostcount and the member linestring carry over between callbacks);
  GPU: urf_marker_points against oracle B, and the C++ adapter's MarkerArray against oracle B.
boost::geometry::simplify is not available here: the stand-in header of oracle A, oracle B and the
product each restate Douglas-Peucker (oracle/urf_rdp.h); that one step is pinned by known answers instead
(tests/test_simplify_kat.py: the worked example of Boost.Geometry's documentation and hand-derived cases)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQ = [("cfg2", 1, 1), ("narrow", 2, 2), ("cfg2", 1, 3), ("narrow", 2, 4)]   # (cfg, scene, seed)


def b_markers(seq, p, mp):
    st = O.OracleMarkerState()
    out = []
    for cfg, _, seed in seq:
        x, y, z = O.cfg_cloud(cfg, seed)
        _, _, dbg = O.run_b(x, y, z, p, debug=True)
        out.append((dbg["marker_pts"], O.marker_strips_b(dbg["marker_pts"], mp, st)))
    return out


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("simp,zavg", [(1, 1), (0, 0), (1, 0), (0, 1)])
def test_oracle_b_markers_equal_reference(simp, zavg):
    mp = O.MarkerParams.default()
    mp.simple_poly_allow, mp.poly_z_avg_allow = simp, zavg
    p = O.cfg_params("cfg2")
    scans = [O.cfg_cloud(cfg, seed) for cfg, _, seed in SEQ]
    _, ia, _, _ = O.run_a(scans, p, marker_params=mp)
    for k, (_, mb) in enumerate(b_markers(SEQ, p, mp)):
        assert ia[k]["markers"] is not None and len(ia[k]["markers"]) >= 3
        assert O.markers_equal(ia[k]["markers"], mb), "sweep %d" % k


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
def test_no_markers_for_degenerate_sweeps():
    """Fewer than 3 marker points: the reference publishes no MarkerArray (lidar_segmentation.cpp:371)."""
    p = O.cfg_params("cfg1")      # z_zero only, no blind spots -> everything road ... but ROI cut to a sliver:
    p.min_Y, p.max_Y = 0.0, 0.2
    x, y, z = O.cfg_cloud("cfg1", 1)
    mp = O.MarkerParams.default()
    _, ia, _, _ = O.run_a([(x, y, z)], p, marker_params=mp)
    _, ib, dbg = O.run_b(x, y, z, p, debug=True)
    mb = O.marker_strips_b(dbg["marker_pts"], mp, O.OracleMarkerState())
    assert O.markers_equal(ia[0]["markers"], mb)


def test_rdp_restatement_basics():
    import ctypes as C
    L = O.oracle_b()
    x = np.array([0, 1, 2, 3, 4, 5], np.float32)
    y = np.array([0, 0.1, 0, 2.0, 0, 0], np.float32)
    keep = np.zeros(6, np.uint8)
    L.urf_rdp_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    L.urf_rdp_float(x.ctypes.data, y.ctypes.data, 6, C.c_float(0.5), keep.ctypes.data)
    assert keep.tolist() == [1, 0, 1, 1, 1, 1] or keep.tolist() == [1, 0, 0, 1, 1, 1] or keep[0] == keep[-1] == keep[3] == 1
    L.urf_rdp_float(x.ctypes.data, y.ctypes.data, 6, C.c_float(5.0), keep.ctypes.data)
    assert keep.tolist() == [1, 0, 0, 0, 0, 1]
    L.urf_rdp_float(x.ctypes.data, y.ctypes.data, 2, C.c_float(5.0), keep.ctypes.data)
    assert keep[:2].tolist() == [1, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,seed,tweak", [("cfg2", 11, {}), ("narrow", 12, {}), ("narrow", 13, {"xDirection": 1}),
                                            ("default_roi", 14, {}), ("cfg5", 2, {}), ("cfg1", 3, {})])
def test_gpu_marker_points(cfg, seed, tweak):
    from golden.make_golden import case_params
    p = case_params(cfg, tweak)
    x, y, z = O.cfg_cloud(cfg, seed)
    _, _, dbg = O.run_b(x, y, z, p, debug=True)
    with u.Context(len(x), 1, params=p) as ctx:
        ctx.classify_xyz(x, y, z)
        got = ctx.marker_points()
    assert got.shape == dbg["marker_pts"].shape and np.array_equal(got, dbg["marker_pts"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(20))
def test_gpu_marker_points_random(seed):
    from fuzz import case
    (x, y, z), p = case(5000 + seed)
    _, ib, dbg = O.run_b(x, y, z, p, debug=True)
    with u.Context(max(len(x), 64), 1, params=p) as ctx:
        ctx.classify_xyz(x, y, z)
        got = ctx.marker_points()
    want = dbg["marker_pts"] if ib["status"] == 0 else np.zeros((0, 4), np.float32)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("simp,zavg", [(1, 1), (0, 0)])
def test_adapter_marker_array(tmp_path, simp, zavg):
    from test_gpu_detector import build_demo   # (same compile line: g++ against the product library)
    exe = build_demo(tmp_path, "marker_demo")
    out = str(tmp_path / "markers.bin")
    files = []
    for k, (_, scene, seed) in enumerate(SEQ):   # the demo links the product library only: the sweeps come from here
        x, y, z = u.synth_cloud(64, 2048, scene, seed)
        files.append(str(tmp_path / ("sweep%d.bin" % k)))
        with open(files[-1], "wb") as f:
            f.write(struct.pack("<I", len(x)) + x.tobytes() + y.tobytes() + z.tobytes())
    r = subprocess.run([exe, str(simp), str(zavg), out] + files, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    blob = open(out, "rb").read()
    mp = O.MarkerParams.default()
    mp.simple_poly_allow, mp.poly_z_avg_allow = simp, zavg
    want = b_markers(SEQ, O.cfg_params("cfg2"), mp)
    pos = 0
    for k in range(len(SEQ)):
        published, nm = struct.unpack_from("<2I", blob, pos)
        pos += 8
        ms = []
        for _ in range(nm):
            mid, act, typ = struct.unpack_from("<3i", blob, pos)
            col = struct.unpack_from("<4f", blob, pos + 12)
            (npt,) = struct.unpack_from("<I", blob, pos + 28)
            pos += 32
            pts = np.frombuffer(blob, np.float64, 3 * npt, pos).reshape(-1, 3).copy()
            pos += 24 * npt
            ms.append({"id": mid, "action": act, "type": typ, "color": col, "points": pts})
        assert O.markers_equal(ms if published else None, want[k][1]), "sweep %d" % k
