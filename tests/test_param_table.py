"""The live parameter surface: urf_param_table() must be the reference's
dynamic_reconfigure description, cfg/LidarFilters.cfg:10-84, row by row -- parsed from the reference
where it is mounted, from the committed fixture (tests/golden/make_cfg_fixture.py) elsewhere -- and
urf_clamp_params() must do what the dynamic_reconfigure server does to a request."""
import ctypes
import json
import math
import os

import pytest

import urban_road_filter_amd as u
from golden.make_cfg_fixture import DEFAULT, parse_cfg

HERE = os.path.dirname(os.path.abspath(__file__))
TYPES = {"bool_t": u.PARAM_BOOL, "int_t": u.PARAM_INT, "double_t": u.PARAM_DOUBLE, "str_t": u.PARAM_STR}


def cfg_rows():
    if os.path.exists(DEFAULT):
        return parse_cfg(DEFAULT)
    return json.load(open(os.path.join(HERE, "golden", "lidar_filters_cfg.json")))


def test_fixture_is_the_reference_cfg():
    if not os.path.exists(DEFAULT):
        pytest.skip("/root/reference is not mounted")
    fixture = json.load(open(os.path.join(HERE, "golden", "lidar_filters_cfg.json")))
    live = json.loads(json.dumps(parse_cfg(DEFAULT)))   # tuples -> lists, as in the file
    assert fixture == live


def test_table_equals_cfg():
    rows, table = cfg_rows(), u.param_table()
    assert len(table) == len(rows) == 28
    for r, t in zip(rows, table):
        assert t["cfg_name"] == r["name"] and t["cfg_line"] == r["line"] and t["type"] == TYPES[r["type"]], r["name"]
        if r["type"] == "str_t":
            assert t["def_str"] == r["default"] and t["where"] == 2
            continue
        assert t["default"] == float(r["default"]), r["name"]
        lo, hi = (0, 1) if r["type"] == "bool_t" else (r["min"], r["max"])
        assert (t["min"], t["max"]) == (float(lo), float(hi)), r["name"]
        if r["enum"]:
            assert t["enum_values"] == ",".join("%s=%d" % (n, v) for n, v in r["enum"])
        else:
            assert t["enum_values"] is None


def test_table_points_at_the_struct_members_and_their_defaults():
    p, mp = u.default_params(), u.default_marker_params()
    for t in u.param_table():
        if t["where"] == 2:
            continue
        struct = p if t["where"] == 0 else mp
        fld = dict((n, (getattr(type(struct), n).offset, ty)) for n, ty in struct._fields_)
        assert t["field"] in fld, t["field"]
        off, cty = fld[t["field"]]
        assert off == t["offset"]
        assert cty is (ctypes.c_float if t["type"] == u.PARAM_DOUBLE else ctypes.c_int32)
        want = ctypes.c_float(t["default"]).value if t["type"] == u.PARAM_DOUBLE else int(t["default"])
        assert getattr(struct, t["field"]) == want, t["cfg_name"]       # urf_default_params == the cfg defaults
    hot = {t["field"] for t in u.param_table() if t["where"] == 0}
    assert hot == {n for n, _ in u.Params._fields_} - {"size", "channels", "sectors", "beam_width"}


def test_clamp_is_dynamic_reconfigures_clamp():
    p, mp = u.default_params(), u.default_marker_params()
    assert u.clamp_params(p, mp) == 0                                    # defaults are inside their ranges
    p.interval, p.curbPoints, p.beamZone, p.min_X, p.xDirection, p.x_zero_method = 1e-4, 99, 5.0, -1e9, 7, 5
    p.dmin_param, p.kdev_param, p.angleFilter1 = -3, float("nan"), 181.0
    mp.poly_s_param, mp.poly_z_manual = 3.0, -9.0
    assert u.clamp_params(p, mp) == 11
    assert (p.interval, p.curbPoints, p.beamZone, p.min_X, p.xDirection, p.x_zero_method) == (
        ctypes.c_float(0.01).value, 30, 10.0, -200.0, 2, 1)
    assert (p.dmin_param, p.kdev_param, p.angleFilter1) == (3, 0.5, 180.0)
    assert (mp.poly_s_param, mp.poly_z_manual) == (1.0, -5.0)
    assert u.clamp_params(p) == 0 and u.lib().urf_clamp_params(None, None, None) == -1
    assert not math.isnan(p.kdev_param)
