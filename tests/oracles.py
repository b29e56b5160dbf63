"""Test-side access to the two CPU oracles (TEST INFRASTRUCTURE, see oracle/).

oracle B = oracle/liburf_oracle.so   (C restatement, travels to the GPU box)
oracle A = oracle/_ref/urf_ref       (the reference's own sources; built only where
                                      /root/reference exists, the binary travels too)
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

from urban_road_filter_amd.api import Params, ScanInfo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_B = os.path.join(ORACLE_DIR, "liburf_oracle.so")
ORACLE_A = os.path.join(ORACLE_DIR, "_ref", "urf_ref")
ORACLE_A_LIBM = os.path.join(ORACLE_DIR, "_ref", "urf_ref_libm")   # same sources, float acos/asin/atan2 -> include/urf_libm.h
REFERENCE = "/root/reference"


class MarkerParams(C.Structure):
    """struct urf_marker_params (cfg/LidarFilters.cfg:75-84)."""
    _fields_ = [("size", C.c_uint32), ("simple_poly_allow", C.c_int32), ("poly_s_param", C.c_float),
                ("poly_z_manual", C.c_float), ("poly_z_avg_allow", C.c_int32)]

    @classmethod
    def default(cls):
        return cls(C.sizeof(cls), 1, 0.7, -1.5, 1)


class OracleMarker(C.Structure):
    _fields_ = [("id", C.c_int32), ("action", C.c_int32), ("type", C.c_int32), ("r", C.c_float), ("g", C.c_float),
                ("b", C.c_float), ("a", C.c_float), ("first_point", C.c_uint32), ("n_points", C.c_uint32)]


class OracleMarkers(C.Structure):
    _fields_ = [("markers", C.POINTER(OracleMarker)), ("n_markers", C.c_uint32), ("cap_markers", C.c_uint32),
                ("pts", C.POINTER(C.c_double)), ("n_points", C.c_uint32), ("cap_points", C.c_uint32),
                ("published", C.c_int32)]


class OracleMarkerState(C.Structure):
    _fields_ = [("ghostcount", C.c_int32), ("line_x", C.c_float * 1024), ("line_y", C.c_float * 1024),
                ("line_n", C.c_int32)]


def marker_strips_b(marker_pts, mparams, state):
    """oracle B: marker points [k,4] -> list of markers (dicts) or None when nothing is published."""
    L = oracle_b()
    L.urf_oracle_marker_strips.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(MarkerParams), C.POINTER(OracleMarkerState),
                                           C.POINTER(OracleMarkers)]
    L.urf_oracle_marker_strips.restype = C.c_int
    mk = (OracleMarker * 1200)()
    pts = (C.c_double * (3 * 8000))()
    out = OracleMarkers(mk, 0, 1200, pts, 0, 8000, 0)
    mp = np.ascontiguousarray(marker_pts, np.float32).reshape(-1, 4)
    rc = L.urf_oracle_marker_strips(mp.ctypes.data, len(mp), C.byref(mparams), C.byref(state), C.byref(out))
    assert rc == 0, rc
    if not out.published:
        return None
    res = []
    for i in range(out.n_markers):
        m = mk[i]
        p = np.array(pts[3 * m.first_point:3 * (m.first_point + m.n_points)], np.float64).reshape(-1, 3)
        res.append({"id": m.id, "action": m.action, "type": m.type, "color": (m.r, m.g, m.b, m.a), "points": p})
    return res


def markers_equal(ma, mb):
    if ma is None or mb is None:
        return ma is None and mb is None
    if len(ma) != len(mb):
        return False
    for a, b in zip(ma, mb):
        if (a["id"], a["action"], a["type"]) != (b["id"], b["action"], b["type"]):
            return False
        if tuple(np.float32(a["color"])) != tuple(np.float32(b["color"])):
            return False
        if a["points"].shape != b["points"].shape or not np.array_equal(a["points"], b["points"]):
            return False
    return True


class OracleDebug(C.Structure):
    _fields_ = [("valpha", C.c_void_p), ("ring", C.c_void_p), ("azimuth", C.c_void_p),
                ("range2d", C.c_void_p), ("detect", C.c_void_p), ("sector", C.c_void_p),
                ("angle_table", C.c_void_p), ("max_dist", C.c_void_p), ("quadrants", C.c_void_p),
                ("beam_stop", C.c_void_p), ("road_order", C.c_void_p), ("curb_order", C.c_void_p),
                ("ring10_order", C.c_void_p), ("marker_pts", C.c_void_p), ("n_marker_pts", C.c_void_p)]


_B = None


def ensure_built():
    """Builds what can be built here (oracle B always; oracle A when the reference is mounted)."""
    need_b = not os.path.exists(ORACLE_B)
    need_a = os.path.isdir(REFERENCE) and not os.path.exists(ORACLE_A)
    if need_b or need_a:
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def oracle_b():
    global _B
    if _B is None:
        ensure_built()
        L = C.CDLL(ORACLE_B)
        L.urf_oracle_classify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(Params),
                                          C.c_void_p, C.POINTER(ScanInfo), C.POINTER(OracleDebug)]
        L.urf_oracle_classify.restype = C.c_int
        for f in ("urf_oracle_acosf", "urf_oracle_asinf"):
            getattr(L, f).argtypes = [C.c_float]
            getattr(L, f).restype = C.c_float
        L.urf_oracle_atan2f.argtypes = [C.c_float, C.c_float]
        L.urf_oracle_atan2f.restype = C.c_float
        _B = L
    return _B


def has_oracle_a():
    ensure_built()
    return os.path.exists(ORACLE_A)


def run_b(x, y, z, params, debug=False):
    """Returns (labels uint8[n], info dict, stages dict or None)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    z = np.ascontiguousarray(z, np.float32)
    n = len(x)
    labels = np.zeros(n, np.uint8)
    info = ScanInfo()
    dbg = None
    st = None
    if debug:
        ch = params.channels
        st = {"valpha": np.zeros(n, np.float32), "ring": np.zeros(n, np.int16),
              "azimuth": np.zeros(n, np.float32), "range2d": np.zeros(n, np.float32),
              "detect": np.zeros(n, np.uint8), "sector": np.zeros(n, np.int16),
              "angle_table": np.zeros(ch, np.float32), "max_dist": np.zeros(ch, np.float32),
              "quadrants": np.zeros(4, np.float32), "beam_stop": np.zeros(2 * 361, np.int16),
              "road_order": np.zeros(n, np.uint32), "curb_order": np.zeros(n, np.uint32),
              "ring10_order": np.zeros(n, np.uint32), "marker_pts": np.zeros(361 * 4, np.float32),
              "n_marker_pts": np.zeros(1, np.uint32)}
        dbg = OracleDebug(*[st[k].ctypes.data for k, _ in OracleDebug._fields_])
    rc = oracle_b().urf_oracle_classify(x.ctypes.data, y.ctypes.data, z.ctypes.data, n, C.byref(params),
                                        labels.ctypes.data, C.byref(info), C.byref(dbg) if dbg else None)
    if rc < 0:
        raise RuntimeError("oracle B failed: %d" % rc)
    if st is not None:
        st["road_order"] = st["road_order"][:info.n_road]
        st["curb_order"] = st["curb_order"][:info.n_curb]
        st["ring10_order"] = st["ring10_order"][:info.n_ring10]
        st["marker_pts"] = st["marker_pts"][:4 * int(st["n_marker_pts"][0])].reshape(-1, 4)
    return labels, info.as_dict(), st


def run_a(scans, params, repeat=1, timeout=1200, marker_params=None, libm=False):
    """scans: list of (x, y, z) with equal length.  Returns (list of labels, list of info dicts,
    ms_per_scan_steady, ms_first).  The RING bit is not observable from the reference.
    With marker_params the info dicts also carry "markers": the road_marker MarkerArray of the scan
    (list of dicts) or None when the reference published none; the scans run in sequence in one
    Detector, as in the node.
    libm=True runs the variant whose float acos / asin / atan2 calls are mapped onto include/urf_libm.h
    (oracle/shim/urf_libm_override.h): same reference sources, the product's definition of the three
    functions instead of the host's glibc."""
    assert has_oracle_a(), "oracle A (reference build) not available"
    n = len(scans[0][0])
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(b"URFREFIN")
            f.write(struct.pack("<4I", len(scans), n, repeat, 1 if marker_params is not None else 0))
            f.write(bytes(params))
            if marker_params is not None:
                f.write(bytes(marker_params))
            for x, y, z in scans:
                assert len(x) == n
                f.write(np.ascontiguousarray(x, np.float32).tobytes())
                f.write(np.ascontiguousarray(y, np.float32).tobytes())
                f.write(np.ascontiguousarray(z, np.float32).tobytes())
        subprocess.run([ORACLE_A_LIBM if libm else ORACLE_A, fin, fout], check=True, timeout=timeout)
        with open(fout, "rb") as f:
            blob = f.read()
    assert blob[:8] == b"URFREFOU"
    ns, nn = struct.unpack_from("<2I", blob, 8)
    ms_steady, ms_first = struct.unpack_from("<2d", blob, 16)
    pos = 32
    labels, infos = [], []
    for _ in range(ns):
        info = ScanInfo.from_buffer_copy(blob[pos:pos + C.sizeof(ScanInfo)])
        pos += C.sizeof(ScanInfo)
        labels.append(np.frombuffer(blob, np.uint8, nn, pos).copy())
        pos += nn
        d = info.as_dict()
        for key, cnt in (("road_order", info.n_road), ("curb_order", info.n_curb), ("ring10_order", info.n_ring10)):
            d[key] = np.frombuffer(blob, np.uint32, cnt, pos).copy()   # the published order of the reference
            pos += 4 * cnt
        if marker_params is not None:
            published, nm = struct.unpack_from("<2I", blob, pos)
            pos += 8
            ms = []
            for _ in range(nm):
                mid, act, typ = struct.unpack_from("<3i", blob, pos)
                col = struct.unpack_from("<4f", blob, pos + 12)
                (npt,) = struct.unpack_from("<I", blob, pos + 28)
                pos += 32
                p = np.frombuffer(blob, np.float64, 3 * npt, pos).reshape(-1, 3).copy()
                pos += 24 * npt
                ms.append({"id": mid, "action": act, "type": typ, "color": col, "points": p})
            d["markers"] = ms if published else None
        infos.append(d)
    return labels, infos, ms_steady, ms_first


# ---- workload configurations (BASELINE.json configs / SURVEY.md section 8d) ------------------
def cfg_params(name):
    """Parameter sets of the named configurations."""
    from urban_road_filter_amd import default_params
    p = default_params().wide_roi()
    if name == "cfg1":      # 16x1024 flat, z_zero only
        p.x_zero_method, p.star_shaped_method, p.blind_spots = 0, 0, 0
    elif name in ("cfg2", "narrow", "sensor", "sensor_narrow"):    # 64x2048 street, all detectors + blind_spots
        pass
    elif name in ("cfg5", "sensor5"):    # 128x4096, channels 128, interval 0.05
        p.channels, p.interval = 128, 0.05
    elif name == "sensor_default_roi":
        p = default_params()
    elif name == "default_roi":
        p = default_params()
    elif name in ("boundary", "boundary_hi"):   # boundary_cloud(): points on the decisions of the float fast paths
        sc = BOUNDARY_SCALE[name]
        p = default_params()
        p.min_X, p.max_X, p.min_Y, p.max_Y, p.min_Z, p.max_Z = -60 * sc, 60 * sc, -60 * sc, 60 * sc, -3 * sc, -1 * sc
        p.channels = 32
    else:
        raise KeyError(name)
    return p


def cfg_cloud(name, seed=1):
    from urban_road_filter_amd import synth_cloud
    if name == "cfg1":
        return synth_cloud(16, 1024, 0, seed)
    if name in ("cfg2", "default_roi"):
        return synth_cloud(64, 2048, 1, seed)
    if name == "narrow":   # curbs within reach of ring 1: blind-spot quadrants engage
        return synth_cloud(64, 2048, 2, seed)
    if name == "cfg5":
        return synth_cloud(128, 4096, 1, seed)
    # sensor-like sweeps (range noise, 2 mm range steps, drop-outs, ~10 000 planar-range ties per 64 x 2048 sweep)
    if name in ("sensor", "sensor_default_roi"):
        return synth_cloud(64, 2048, 3, seed)
    if name == "sensor_narrow":
        return synth_cloud(64, 2048, 4, seed)
    if name == "sensor5":
        return synth_cloud(128, 4096, 3, seed)
    if name in BOUNDARY_SCALE:
        return boundary_cloud(BOUNDARY_SCALE[name], seed)
    raise KeyError(name)


def case_cloud(name, seed, fixture=None):
    """cfg_cloud, except that clouds built with numpy transcendentals (boundary*) are taken from
    the golden fixture that stores them: their last bits may depend on the numpy build."""
    if fixture is not None and "x" in fixture:
        return fixture["x"], fixture["y"], fixture["z"]
    return cfg_cloud(name, seed)


BOUNDARY_SCALE = {"boundary": 1.0, "boundary_hi": 2.0 ** 30}


def boundary_cloud(scale=1.0, seed=3):
    """Points placed ON the decisions the float fast paths take: vertical angles at a ring-table
    entry +- interval (in steps of the float resolution, out to beyond the fast path's margin), polar
    angles at integer sector boundaries +- 0 .. 3e-4 deg, hence azimuths at integer degrees too."""
    rng = np.random.default_rng(seed)
    h = 1.8
    lead = 62.0 + 1.7 * np.arange(14)                                   # table entries [deg from -z]
    va = [lead]
    offs = np.concatenate([np.arange(-120, 121) * 2.0e-6, [-3.2e-4, -3.0e-4, -2.8e-4, 2.8e-4, 3.0e-4, 3.2e-4]])
    for sgn in (-1.0, 1.0):
        for l in lead[::3]:
            va.append(l + sgn * 0.18 + offs)
    va = np.concatenate(va)
    fi = rng.uniform(3.0, 357.0, len(va))
    # sector / azimuth boundaries: every integer degree of a few decades, tiny offsets either side
    deg = np.arange(1, 360, 7, dtype=np.float64)
    d = np.array([0.0, 1e-7, 1e-6, 1e-5, 5e-5, 1e-4, 2e-4, 2.4e-4, 2.6e-4, 3e-4])
    fb = (deg[:, None] + np.concatenate([-d[1:], d])[None, :]).ravel()
    vb = lead[rng.integers(0, len(lead), len(fb))] + rng.uniform(-0.1, 0.1, len(fb))
    va = np.concatenate([va, vb])
    fi = np.concatenate([fi, fb])
    rho = h * np.tan(np.deg2rad(va))
    rho *= 1.0 + 1e-4 * np.arange(len(rho)) / len(rho)                   # no planar-range ties (and rings stay put)
    z = np.full(len(rho), -h) * (1.0 + 1e-4 * np.arange(len(rho)) / len(rho))
    x, y = rho * np.cos(np.deg2rad(fi)), rho * np.sin(np.deg2rad(fi))
    return (x * scale).astype(np.float32), (y * scale).astype(np.float32), (z * scale).astype(np.float32)





MASK_NO_RING = 0xFF & ~0x08   # oracle A cannot report the RING bit
