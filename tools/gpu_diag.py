#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP path with oracle B on one configuration.
Diagnostic tool for bring-up (tests/ holds the asserting versions).
    python tools/gpu_diag.py [cfg1|cfg2|cfg5|default_roi] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracles as O  # noqa: E402
import urban_road_filter_amd as u  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    p = O.cfg_params(cfg)
    x, y, z = O.cfg_cloud(cfg, seed)
    n = len(x)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx = u.Context(n, 1, params=p)
    ctx.enable_stage_capture(True)
    t = time.time()
    lg, ig = ctx.classify_xyz(x, y, z)
    print("gpu call %.1f ms; info gpu %s" % ((time.time() - t) * 1e3, ig.as_dict()))
    print("info oracle  %s" % ib)

    def cmp(name, a, b, mask=None, exact=True):
        if mask is not None:
            a, b = a[mask], b[mask]
        if exact:
            bad = np.nonzero(a != b)[0]
        else:
            bad = np.nonzero(~(np.abs(a - b) <= 1e-5 * np.maximum(1, np.abs(b))))[0]
        bits = ""
        if a.dtype == np.float32 and len(a):
            nbit = (a.view(np.uint32) != b.view(np.uint32)).sum()
            bits = " (bit-different: %d)" % nbit
        print("%-12s n=%d mismatches=%d%s" % (name, len(a), len(bad), bits))
        if len(bad):
            i = bad[:8]
            print("    idx", i, "gpu", a[i], "oracle", b[i])
        return len(bad)

    roi = (lb & u.FLAG_ROI) != 0
    ring = (lb & u.FLAG_RING) != 0
    cmp("valpha", ctx.read_stage(u.STAGE_VALPHA, n), st["valpha"], roi)
    cmp("angle_table", ctx.read_stage(u.STAGE_ANGLE_TABLE, n), st["angle_table"])
    cmp("ring", ctx.read_stage(u.STAGE_RING, n), st["ring"])
    if p.star_shaped_method:
        cmp("sector", ctx.read_stage(u.STAGE_SECTOR, n), st["sector"])
    cmp("azimuth", ctx.read_stage(u.STAGE_AZIMUTH, n), st["azimuth"], ring)
    cmp("range2d", ctx.read_stage(u.STAGE_RANGE2D, n), st["range2d"], ring)
    cmp("detect", ctx.read_stage(u.STAGE_DETECT, n), st["detect"])
    cmp("max_dist", ctx.read_stage(u.STAGE_MAXDIST, n), st["max_dist"])
    cmp("quadrants", ctx.read_stage(u.STAGE_QUADRANTS, n), st["quadrants"])
    cmp("beam_stop", ctx.read_stage(u.STAGE_BEAM_STOP, n), st["beam_stop"])
    bad = cmp("labels", lg, lb)
    for k in ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10"):
        if getattr(ig, k) != ib[k]:
            print("info.%s differs: gpu %d oracle %d" % (k, getattr(ig, k), ib[k]))
            bad += 1
    print("RESULT", cfg, seed, "OK" if bad == 0 else "MISMATCH")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
