"""ctypes binding of include/urf.h (and, for tests and bench.py, of include/urf_test_hooks.h).  No compute
happens in Python and there is no fallback: if ``liburf_hip.so`` is missing the import of :func:`lib` raises.

Two libraries: ``liburf_hip.so`` is the product (exactly include/urf.h); ``liburf_hip_test.so`` is the same
sources plus the test / benchmark hooks.  ``Context(...)`` lives in the product library, ``Context(..., hooks=True)``
in the hooks build (needed for set_debug_flags / selftest* / bench_callback_stream)."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

LABEL_MASK, LABEL_ROAD, LABEL_CURB = 0x03, 1, 2
FLAG_ROI, FLAG_RING, FLAG_RING10 = 0x04, 0x08, 0x10
(STAGE_VALPHA, STAGE_RING, STAGE_AZIMUTH, STAGE_RANGE2D, STAGE_DETECT, STAGE_SECTOR,
 STAGE_ANGLE_TABLE, STAGE_MAXDIST, STAGE_QUADRANTS, STAGE_BEAM_STOP) = range(1, 11)

OK, TOO_FEW_POINTS = 0, 1


class UrfError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("urf error %d (%s) %s" % (code, _strerror(code), what))


class Params(C.Structure):
    """struct urf_params (include/urf.h); field names follow the reference's namespace params."""
    _fields_ = [
        ("size", C.c_uint32),
        ("x_zero_method", C.c_int32), ("z_zero_method", C.c_int32),
        ("star_shaped_method", C.c_int32), ("blind_spots", C.c_int32),
        ("xDirection", C.c_int32),
        ("interval", C.c_float), ("curbHeight", C.c_float),
        ("curbPoints", C.c_int32),
        ("beamZone", C.c_float),
        ("min_X", C.c_float), ("max_X", C.c_float),
        ("min_Y", C.c_float), ("max_Y", C.c_float),
        ("min_Z", C.c_float), ("max_Z", C.c_float),
        ("angleFilter1", C.c_float), ("angleFilter2", C.c_float), ("angleFilter3", C.c_float),
        ("kdev_param", C.c_float), ("kdist_param", C.c_float),
        ("starbeam_filter", C.c_int32), ("dmin_param", C.c_int32),
        ("channels", C.c_int32), ("sectors", C.c_int32),
        ("beam_width", C.c_float),
    ]

    def copy(self):
        p = Params()
        C.memmove(C.byref(p), C.byref(self), C.sizeof(Params))
        return p

    def wide_roi(self, half=200.0):
        """ROI x,y widened to +-half metres (so that every return of a
        full sweep is classified); z keeps the reference default [-3, -1]."""
        self.min_X, self.max_X, self.min_Y, self.max_Y = -half, half, -half, half
        return self


class ScanInfo(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_roi", C.c_uint32), ("n_rings", C.c_uint32),
                ("n_ring_pts", C.c_uint32), ("n_road", C.c_uint32), ("n_curb", C.c_uint32),
                ("n_ring10", C.c_uint32), ("n_nan_azimuth", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class MarkerParams(C.Structure):
    """struct urf_marker_params: the polygon parameters of the road_marker output."""
    _fields_ = [("size", C.c_uint32), ("simple_poly_allow", C.c_int32), ("poly_s_param", C.c_float),
                ("poly_z_manual", C.c_float), ("poly_z_avg_allow", C.c_int32)]


class ParamDesc(C.Structure):
    """struct urf_param_desc: one row of the reference's dynamic_reconfigure interface."""
    _fields_ = [("cfg_name", C.c_char_p), ("field", C.c_char_p), ("where", C.c_int32), ("offset", C.c_uint32),
                ("type", C.c_int32), ("default", C.c_double), ("min", C.c_double), ("max", C.c_double),
                ("def_str", C.c_char_p), ("enum_values", C.c_char_p), ("cfg_line", C.c_int32)]


PARAM_BOOL, PARAM_INT, PARAM_DOUBLE, PARAM_STR = range(4)

_LIBS = {}

HOOK_SYMBOLS = ("urf_set_debug_flags", "urf_selftest", "urf_selftest_fast", "urf_synth_cloud", "urf_bench_callback_stream")


def lib_path(hooks=False):
    # URF_LIB_PATH: tuning experiments load an alternative build (always one with the hooks) for everything
    return os.environ.get("URF_LIB_PATH") or os.path.join(_HERE, "liburf_hip_test.so" if hooks else "liburf_hip.so")


def test_lib():
    """liburf_hip_test.so: the product's sources plus include/urf_test_hooks.h."""
    return lib(hooks=True)


def lib(hooks=False):
    """Loads liburf_hip.so (hooks=True: liburf_hip_test.so).  When torch is (or will be) in the process it
    must be imported first so that both share ONE HIP runtime (torch bundles libamdhip64.so.7; ours is
    resolved by SONAME to whichever copy is already loaded)."""
    path = lib_path(hooks)
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise ImportError("%s not built: run `python -m urban_road_filter_amd.build`" % path)
    L = C.CDLL(path, mode=C.RTLD_LOCAL)   # (both builds export the same names: each keeps to itself, -Bsymbolic)
    has_hooks = hasattr(L, "urf_synth_cloud")
    vp, u8p, u32p, fp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    sig = {
        "urf_default_params": [C.POINTER(Params)],
        "urf_default_marker_params": [C.POINTER(MarkerParams)],
        "urf_create": [C.POINTER(C.c_void_p), C.c_int, C.c_uint32, C.c_uint32],
        "urf_destroy": [vp],
        "urf_set_params": [vp, C.POINTER(Params)],
        "urf_get_params": [vp, C.POINTER(Params)],
        "urf_set_stream": [vp, vp],
        "urf_synchronize": [vp],
        "urf_classify_pc2": [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.POINTER(ScanInfo)],
        "urf_classify_pc2_async": [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)],
        "urf_classify_pc2_wait": [vp, C.c_uint32, u8p, C.POINTER(ScanInfo)],
        "urf_result_labels": [vp, C.c_uint32, C.POINTER(C.c_void_p)],
        "urf_pinned_input": [vp, C.c_size_t, C.POINTER(C.c_void_p)],
        "urf_param_count": [],
        "urf_clamp_params": [C.POINTER(Params), C.POINTER(MarkerParams), C.POINTER(C.c_uint32)],
        "urf_classify_batch_soa": [vp, fp, fp, fp, C.c_uint32, C.c_uint32, u8p, vp],
        "urf_classify_batch_soa_ragged": [vp, fp, fp, fp, u32p, C.c_uint32, C.c_uint32, u8p, vp],
        "urf_classify_batch_pc2": [vp, u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p, vp],
        "urf_compact_indices": [vp, u8p, C.c_uint32, u32p, u32p, u32p, u32p, u32p],
        "urf_compact_indices_batch": [vp, u8p, C.c_uint32, C.c_uint32, u32p, u32p, u32p, u32p, u32p],
        "urf_ordered_indices_batch": [vp, u32p, u32p, u32p, C.c_uint32, u32p],
        "urf_marker_points_batch": [vp, vp, u32p],
        "urf_read_stage": [vp, C.c_int, C.c_uint32, vp, C.c_size_t],
        "urf_ordered_indices": [vp, C.c_uint32, vp, vp, vp, vp],
        "urf_marker_points": [vp, C.c_uint32, vp, vp],
        "urf_enable_stage_capture": [vp, C.c_int],
        "urf_set_debug_flags": [vp, C.c_uint32],
        "urf_callback_path_state": [vp, C.c_void_p, C.c_void_p],
        "urf_pc2_to_planes": [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, fp, fp, fp],
        "urf_enable_kernel_timing": [vp, C.c_int],
        "urf_selftest": [vp, C.c_void_p],
        "urf_selftest_fast": [vp, C.c_uint64, C.c_void_p],
        "urf_kernel_timing": [vp, C.c_void_p, C.c_void_p],
        "urf_synth_cloud": [C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, fp, fp, fp],
        "urf_bench_callback_stream": [vp, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_uint32, C.c_int, u8p, C.POINTER(C.c_double)],
        "urf_abi_version": [],
        "urf_set_front_mode": [vp, C.c_int],
        "urf_front_scans": [vp, C.c_void_p],
        "urf_callback_path_preset": [vp, C.c_uint32],
    }
    for name, args in sig.items():
        if name in HOOK_SYMBOLS and not has_hooks:
            continue   # the product library exports none of them (tests/test_abi.py)
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    L.urf_param_table.argtypes = []
    L.urf_param_table.restype = C.POINTER(ParamDesc)
    L.urf_strerror.argtypes = [C.c_int]
    L.urf_strerror.restype = C.c_char_p
    L.urf_kernel_name.argtypes = [C.c_int]
    L.urf_kernel_name.restype = C.c_char_p
    L.urf_last_error.argtypes = [vp]
    L.urf_last_error.restype = C.c_char_p
    L.urf_has_hooks = has_hooks
    _LIBS[path] = L
    return L


def _strerror(code):
    try:
        return lib().urf_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "?"


def default_params():
    p = Params()
    rc = lib().urf_default_params(C.byref(p))
    if rc != 0:
        raise UrfError(rc)
    return p


def default_marker_params():
    p = MarkerParams()
    rc = lib().urf_default_marker_params(C.byref(p))
    if rc != 0:
        raise UrfError(rc)
    return p


def param_table():
    """The reference's parameter surface (cfg/LidarFilters.cfg) as a list of dicts."""
    L = lib()
    t = L.urf_param_table()
    rows = []
    for i in range(L.urf_param_count()):
        d = t[i]
        rows.append(dict(cfg_name=d.cfg_name.decode(), field=d.field.decode(), where=d.where, offset=d.offset, type=d.type,
                         default=d.default, min=d.min, max=d.max, def_str=d.def_str.decode() if d.def_str else None,
                         enum_values=d.enum_values.decode() if d.enum_values else None, cfg_line=d.cfg_line))
    return rows


def clamp_params(p, mp=None):
    """dynamic_reconfigure's clamping of a request; returns the number of values it changed."""
    n = C.c_uint32(0)
    rc = lib().urf_clamp_params(C.byref(p), C.byref(mp) if mp is not None else None, C.byref(n))
    if rc != 0:
        raise UrfError(rc, "urf_clamp_params")
    return n.value


def pc2_to_planes(data, n_points, point_step, off_x, off_y, off_z, out=None):
    """x, y, z (float32 arrays) of a PointCloud2-layout message (uint8 array): the host-side gather of the callback
    path by itself (no GPU).  out: optional (x, y, z) arrays to fill."""
    buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    assert buf.size >= n_points * point_step
    x, y, z = out if out is not None else (np.empty(n_points, np.float32) for _ in range(3))
    for a in (x, y, z):   # raw pointers go to C: a float64 or strided array would be filled wrongly or overrun
        assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags.c_contiguous and a.size >= n_points
    rc = lib().urf_pc2_to_planes(buf.ctypes.data, n_points, point_step, off_x, off_y, off_z, x.ctypes.data, y.ctypes.data,
                                 z.ctypes.data)
    if rc != 0:
        raise UrfError(rc, "urf_pc2_to_planes")
    return x, y, z


def synth_cloud(rings, cols, scene=1, seed=1):
    """SURVEY.md section 8d synthetic sweep: returns float32 arrays x, y, z of rings*cols points in
    firing order (idx = col*rings + ring).  scene 0 = flat ground, 1 = street with curbs."""
    n = rings * cols
    x = np.empty(n, np.float32)
    y = np.empty(n, np.float32)
    z = np.empty(n, np.float32)
    rc = test_lib().urf_synth_cloud(rings, cols, scene, seed, x.ctypes.data, y.ctypes.data, z.ctypes.data)
    if rc != 0:
        raise UrfError(rc, "urf_synth_cloud")
    return x, y, z


def _ptr(obj):
    """Device pointer of a torch tensor / int / None; host pointer of a numpy array."""
    if obj is None:
        return None
    if isinstance(obj, int):
        return obj
    if isinstance(obj, np.ndarray):
        return obj.ctypes.data
    if hasattr(obj, "data_ptr"):
        return obj.data_ptr()
    raise TypeError(type(obj))


class Context:
    """One urf_ctx: one device, one stream, all scratch memory."""

    def __init__(self, max_points, max_batch=1, device=0, params=None, hooks=False):
        self._h = C.c_void_p()
        self._lib = lib(hooks)   # hooks=True: the context lives in liburf_hip_test.so (include/urf_test_hooks.h)
        rc = self._lib.urf_create(C.byref(self._h), device, max_points, max_batch)
        if rc != 0:
            self._h = None
            raise UrfError(rc, "urf_create")
        self.max_points, self.max_batch = max_points, max_batch
        if params is not None:
            self.set_params(params)

    def _check(self, rc, what):
        if rc < 0:
            raise UrfError(rc, what + ": " + self._lib.urf_last_error(self._h).decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.urf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_params(self, p):
        self._check(self._lib.urf_set_params(self._h, C.byref(p)), "urf_set_params")

    def get_params(self):
        p = Params()
        self._check(self._lib.urf_get_params(self._h, C.byref(p)), "urf_get_params")
        return p

    def set_stream(self, stream_handle):
        self._check(self._lib.urf_set_stream(self._h, stream_handle), "urf_set_stream")

    def synchronize(self):
        self._check(self._lib.urf_synchronize(self._h), "urf_synchronize")

    def enable_stage_capture(self, mode=1):
        """0 off; 1 exact arithmetic for every point, all stages readable; 2 production decisions with
        the ring / sector of every input point recorded (STAGE_RING, STAGE_SECTOR)."""
        self._check(self._lib.urf_enable_stage_capture(self._h, int(mode)), "urf_enable_stage_capture")

    def callback_path_state(self):
        """(sweeps run again so far, sequence bits: 1 speculative ring table, 2 work-list kernels launched)."""
        n, q = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.urf_callback_path_state(self._h, C.addressof(n), C.addressof(q)), "urf_callback_path_state")
        return n.value, q.value

    def _need_hooks(self, what):
        if not self._lib.urf_has_hooks:
            raise RuntimeError("%s is a test hook (include/urf_test_hooks.h): create the context with hooks=True" % what)

    def set_debug_flags(self, flags):
        self._need_hooks("urf_set_debug_flags")
        self._check(self._lib.urf_set_debug_flags(self._h, int(flags)), "urf_set_debug_flags")

    NUM_KERNELS = 8

    def selftest_fast(self, n_samples=1 << 27):
        """max |approx - exact| of the float fast paths: (vertical angle [deg], polar angle [rad], fi*Kfi, azimuth [deg])."""
        self._need_hooks("urf_selftest_fast")
        err = np.zeros(4, np.float32)
        self._check(self._lib.urf_selftest_fast(self._h, n_samples, err.ctypes.data), "urf_selftest_fast")
        return tuple(float(v) for v in err)

    def selftest(self):
        self._need_hooks("urf_selftest")
        n = C.c_uint64(0)
        self._check(self._lib.urf_selftest(self._h, C.byref(n)), "urf_selftest")
        return n.value

    def set_front_mode(self, mode):
        """The fused front end for batches of sweeps in firing order (include/urf.h): 0 never, 1 batches of >= 32 scans, 2 always."""
        self._check(self._lib.urf_set_front_mode(self._h, int(mode)), "urf_set_front_mode")

    def callback_path_preset(self, sequence_bits):
        """urf_callback_path_preset: 2 work-list kernels | 4 NaN-azimuth rings | 16 std::sort's tie order, ahead of the first sweep that needs them."""
        self._check(self._lib.urf_callback_path_preset(self._h, int(sequence_bits)), "urf_callback_path_preset")

    def front_scans(self):
        """Scans of the last batch call that took the fused front end."""
        n = C.c_uint32(0)
        self._check(self._lib.urf_front_scans(self._h, C.addressof(n)), "urf_front_scans")
        return n.value

    def enable_kernel_timing(self, on=True):
        self._check(self._lib.urf_enable_kernel_timing(self._h, int(on)), "urf_enable_kernel_timing")

    def kernel_timing(self):
        """-> ({kernel name: summed ms}, number of classify calls) since the last query."""
        ms = (C.c_double * self.NUM_KERNELS)()
        n = C.c_uint32(0)
        self._check(self._lib.urf_kernel_timing(self._h, ms, C.byref(n)), "urf_kernel_timing")
        names = [self._lib.urf_kernel_name(i).decode() for i in range(self.NUM_KERNELS)]
        return dict(zip(names, list(ms))), n.value

    # -- single scan, asynchronous: URF_MAX_IN_FLIGHT = 4 slots, slot i on scratch row i % min(max_batch, 4) and on that
    # row's own stream (copy in, kernels, copy out), so that the sweeps in flight overlap ----
    def classify_pc2_async(self, data, n_points, point_step, off_x, off_y, off_z):
        """data: uint8 array (or the int address urf_pinned_input returned).  Returns a ticket."""
        ptr = data if isinstance(data, int) else np.ascontiguousarray(data).view(np.uint8).ctypes.data
        t = C.c_uint32(0)
        self._check(self._lib.urf_classify_pc2_async(self._h, ptr, n_points, point_step, off_x, off_y, off_z, C.byref(t)),
                    "urf_classify_pc2_async")
        return t.value

    def classify_pc2_wait(self, ticket, labels=None):
        """Blocks until the sweep is done; labels: optional uint8 array to receive the label bytes."""
        info = ScanInfo()
        self._check(self._lib.urf_classify_pc2_wait(self._h, ticket, labels.ctypes.data if labels is not None else None,
                                                    C.byref(info)), "urf_classify_pc2_wait")
        return info

    def bench_callback_stream(self, msgs, n_points, point_step, off_x, off_y, off_z, n_sweeps, in_flight, producer_pinned=False):
        """Seconds the library's own submit / collect loop takes for n_sweeps messages (uint8 arrays), in_flight at a time."""
        self._need_hooks("urf_bench_callback_stream")
        for m in msgs:   # raw pointers go to C
            assert isinstance(m, np.ndarray) and m.flags.c_contiguous and m.nbytes >= n_points * point_step
        arr = (C.c_void_p * len(msgs))(*[m.ctypes.data for m in msgs])
        lab = np.empty(n_points, np.uint8)
        sec = C.c_double(0.0)
        self._check(self._lib.urf_bench_callback_stream(self._h, arr, len(msgs), n_points, point_step, off_x, off_y, off_z,
                                                        n_sweeps, in_flight, 1 if producer_pinned else 0, lab.ctypes.data,
                                                        C.byref(sec)), "urf_bench_callback_stream")
        return sec.value, lab

    def pinned_input(self, nbytes):
        """uint8 numpy view (nbytes long) of the pinned input buffer the NEXT submission will use."""
        p = C.c_void_p()
        self._check(self._lib.urf_pinned_input(self._h, nbytes, C.byref(p)), "urf_pinned_input")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def result_labels(self, ticket, n_points):
        p = C.c_void_p()
        self._check(self._lib.urf_result_labels(self._h, ticket, C.byref(p)), "urf_result_labels")
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n_points,))

    # -- single scan, host, PointCloud2 layout ------------------------------------
    def classify_pc2(self, data, n_points, point_step, off_x, off_y, off_z):
        """data: bytes-like / uint8 array of n_points*point_step bytes.  Returns (labels, ScanInfo)."""
        buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
        assert buf.size >= n_points * point_step
        buf = np.ascontiguousarray(buf)
        labels = np.zeros(n_points, np.uint8)
        info = ScanInfo()
        self._check(self._lib.urf_classify_pc2(self._h, buf.ctypes.data, n_points, point_step, off_x, off_y, off_z,
                                               labels.ctypes.data, C.byref(info)), "urf_classify_pc2")
        return labels, info

    def classify_xyz(self, x, y, z):
        """Convenience: one scan given as three float32 arrays -> packs x,y,z,intensity records
        (the pcl::PointXYZI wire layout, point_step 16) and calls classify_pc2."""
        n = len(x)
        rec = np.zeros((n, 4), np.float32)
        rec[:, 0], rec[:, 1], rec[:, 2] = x, y, z
        return self.classify_pc2(rec, n, 16, 0, 4, 8)

    # -- batches, device resident -------------------------------------------------
    def classify_batch_soa(self, d_x, d_y, d_z, n_per_scan, n_scans, d_labels, d_info=None):
        self._check(self._lib.urf_classify_batch_soa(self._h, _ptr(d_x), _ptr(d_y), _ptr(d_z), n_per_scan, n_scans,
                                                     _ptr(d_labels), _ptr(d_info)), "urf_classify_batch_soa")

    def classify_batch_soa_ragged(self, d_x, d_y, d_z, d_offsets, max_len, n_scans, d_labels, d_info=None):
        self._check(self._lib.urf_classify_batch_soa_ragged(self._h, _ptr(d_x), _ptr(d_y), _ptr(d_z), _ptr(d_offsets),
                                                            max_len, n_scans, _ptr(d_labels), _ptr(d_info)),
                    "urf_classify_batch_soa_ragged")

    def classify_batch_pc2(self, d_data, n_per_scan, n_scans, point_step, off_x, off_y, off_z, d_labels, d_info=None):
        self._check(self._lib.urf_classify_batch_pc2(self._h, _ptr(d_data), n_per_scan, n_scans, point_step,
                                                     off_x, off_y, off_z, _ptr(d_labels), _ptr(d_info)),
                    "urf_classify_batch_pc2")

    def compact_indices(self, d_labels, n_points, d_road, d_curb, d_roi, d_ring10, d_counts):
        self._check(self._lib.urf_compact_indices(self._h, _ptr(d_labels), n_points, _ptr(d_road), _ptr(d_curb),
                                                  _ptr(d_roi), _ptr(d_ring10), _ptr(d_counts)), "urf_compact_indices")

    def compact_indices_batch(self, d_labels, n_per_scan, n_scans, d_road, d_curb, d_roi, d_ring10, d_counts):
        self._check(self._lib.urf_compact_indices_batch(self._h, _ptr(d_labels), n_per_scan, n_scans, _ptr(d_road), _ptr(d_curb),
                                                        _ptr(d_roi), _ptr(d_ring10), _ptr(d_counts)), "urf_compact_indices_batch")

    def ordered_indices_batch(self, d_road, d_curb, d_ring10, stride, d_counts):
        self._check(self._lib.urf_ordered_indices_batch(self._h, _ptr(d_road), _ptr(d_curb), _ptr(d_ring10), stride, _ptr(d_counts)),
                    "urf_ordered_indices_batch")

    def marker_points_batch(self, d_pts, d_counts):
        self._check(self._lib.urf_marker_points_batch(self._h, _ptr(d_pts), _ptr(d_counts)), "urf_marker_points_batch")

    def ordered_indices(self, n_points, scan=0):
        """Input indices of the road / curb / road_probably clouds of scan `scan` in the order the
        reference publishes them (ring-major, azimuth ascending).  Returns three uint32 arrays."""
        bufs = [np.zeros(n_points, np.uint32) for _ in range(3)]
        cnt = np.zeros(3, np.uint32)
        self._check(self._lib.urf_ordered_indices(self._h, scan, bufs[0].ctypes.data, bufs[1].ctypes.data,
                                                  bufs[2].ctypes.data, cnt.ctypes.data), "urf_ordered_indices")
        return tuple(b[:int(c)] for b, c in zip(bufs, cnt))

    def marker_points(self, scan=0):
        """lidar_segmentation.cpp:295-351: the marker points of scan `scan`, float32 [k, 4] = x, y, z, red."""
        buf = np.zeros(361 * 4, np.float32)
        n = C.c_uint32(0)
        self._check(self._lib.urf_marker_points(self._h, scan, buf.ctypes.data, C.byref(n)), "urf_marker_points")
        return buf[:4 * n.value].reshape(-1, 4)

    # -- stage-wise inspection ----------------------------------------------------
    _STAGE_DTYPE = {STAGE_VALPHA: np.float32, STAGE_RING: np.int16, STAGE_AZIMUTH: np.float32,
                    STAGE_RANGE2D: np.float32, STAGE_DETECT: np.uint8, STAGE_SECTOR: np.int16,
                    STAGE_ANGLE_TABLE: np.float32, STAGE_MAXDIST: np.float32, STAGE_QUADRANTS: np.float32,
                    STAGE_BEAM_STOP: np.int16}

    def read_stage(self, what, n_points, scan=0):
        if what in (STAGE_ANGLE_TABLE, STAGE_MAXDIST):
            count = self.get_params().channels
        elif what == STAGE_QUADRANTS:
            count = 4
        elif what == STAGE_BEAM_STOP:
            count = 2 * 361
        else:
            count = n_points
        out = np.zeros(count, self._STAGE_DTYPE[what])
        self._check(self._lib.urf_read_stage(self._h, what, scan, out.ctypes.data, out.nbytes), "urf_read_stage")
        return out
