#!/usr/bin/env python3
"""Generates tests/golden/*.npz from ORACLE A, i.e. from the reference's own unmodified
sources (oracle/Makefile compiles them from /root/reference against oracle/shim).

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py

Each file holds, for one configuration and seed:
    labels      uint8[n]  label byte per input point as the reference publishes it
                (bits: 0-1 isCurbPoint, 0x04 roi, 0x10 road_probably; the RING bit 0x08 is not
                observable from the reference's outputs and is always 0 here)
    info        n_roi, n_road, n_curb, n_ring10, status of the reference
    cloud_sha   sha256 of the x|y|z float32 bytes the labels belong to (guards the generator)
    params      the urf_params bytes used
The reference has no tests or golden vectors of its own; these files are the
pin for oracle B and, through it, for the HIP path.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracles as O  # noqa: E402

# (name, cfg, seed, parameter tweak)
CASES = [
    ("cfg1_s1", "cfg1", 1, {}),
    ("cfg2_s1", "cfg2", 1, {}),
    ("cfg2_s2", "cfg2", 2, {}),
    ("cfg2_s3", "cfg2", 3, {}),
    ("narrow_s1", "narrow", 1, {}),
    ("narrow_s1_xdir1", "narrow", 1, {"xDirection": 1}),
    ("narrow_s1_xdir2", "narrow", 1, {"xDirection": 2}),
    ("narrow_s1_noblind", "narrow", 1, {"blind_spots": 0}),
    ("cfg2_s1_starbeam", "cfg2", 1, {"starbeam_filter": 1}),
    ("cfg2_s1_cp9_bz45", "cfg2", 1, {"curbPoints": 9, "beamZone": 45.5}),
    ("default_roi_s1", "default_roi", 1, {}),
    ("cfg5_s1", "cfg5", 1, {}),
    # points on the ring / sector / integer-degree decisions (tests/oracles.py boundary_cloud); the cloud
    # itself is stored in the fixture
    ("boundary_s3", "boundary", 3, {}),
    ("boundary_s3_starbeam_xdir1", "boundary", 3, {"starbeam_filter": 1, "xDirection": 1}),
    ("boundary_s3_ch20", "boundary", 3, {"channels": 20}),
    ("boundary_hi_s3", "boundary_hi", 3, {}),
    # sensor-like sweeps: the planar-range ties of a real driver's output left in (r5; the order of equal ranges is
    # std::sort's and decides several hundred labels per sweep)
    ("sensor_s1", "sensor", 1, {}),
    ("sensor_s2", "sensor", 2, {}),
    ("sensor_narrow_s1", "sensor_narrow", 1, {}),
    ("sensor_narrow_s1_starbeam_xdir2", "sensor_narrow", 1, {"starbeam_filter": 1, "xDirection": 2}),
    ("sensor_default_roi_s3", "sensor_default_roi", 3, {}),
    ("sensor5_s1", "sensor5", 1, {}),
]


def case_params(cfg, tweak):
    p = O.cfg_params(cfg)
    for k, v in tweak.items():
        setattr(p, k, v)
    return p


def cloud_sha(x, y, z):
    h = hashlib.sha256()
    for a in (x, y, z):
        h.update(np.ascontiguousarray(a, np.float32).tobytes())
    return h.hexdigest()


def main():
    only = sys.argv[1:]
    for name, cfg, seed, tweak in CASES:
        if only and not any(name.startswith(o) for o in only):
            continue
        p = case_params(cfg, tweak)
        x, y, z = O.cfg_cloud(cfg, seed)
        # boundary clouds sit on decisions that one ulp of acosf / asinf / atan2f flips, so their
        # goldens come from the reference sources with the product's libm definition (oracle/shim/
        # urf_libm_override.h); with the host's glibc 2.35 the reference labels 3 of the 3472
        # points of boundary_s3 differently (asinf(0.8660254f) is 1 ulp high there: azimuth 60
        # instead of 59.999996)
        libm = cfg in O.BOUNDARY_SCALE
        labels, infos, ms, _ = O.run_a([(x, y, z)], p, libm=libm)
        info = infos[0]
        out = os.path.join(HERE, name + ".npz")
        extra = {"x": x, "y": y, "z": z, "oracle": "urf_ref_libm"} if libm else {}
        np.savez_compressed(out, labels=labels[0], cloud_sha=cloud_sha(x, y, z), params=np.frombuffer(bytes(p), np.uint8),
                            **{"info_" + k: info[k] for k in ("status", "n_roi", "n_road", "n_curb", "n_ring10")}, **extra)
        print("%-20s n=%d road=%d curb=%d roi=%d  labels sha256 %s  (%d bytes)" % (
            name, len(x), info["n_road"], info["n_curb"], info["n_roi"],
            hashlib.sha256(labels[0].tobytes()).hexdigest()[:16], os.path.getsize(out)))


if __name__ == "__main__":
    main()
