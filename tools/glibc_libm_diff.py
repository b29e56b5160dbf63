#!/usr/bin/env python3
"""How often do the host's glibc and include/urf_libm.h lead the REFERENCE to different labels?  (r5 review, missing #5 / next #7)

The product's float acos / asin / atan2 are defined by include/urf_libm.h (correctly rounded); the reference calls the host's glibc, whose
three functions are not correctly rounded (glibc 2.35: asinf(0.8660254f) comes out 1 ulp high).  On an input that lies ON a decision
boundary -- an azimuth at an integer degree, a vertical angle at a ring window's edge -- one ulp decides a label.  This script runs the
reference's own sources twice on the same sweeps, once linked against glibc (oracle/_ref/urf_ref) and once with the three functions
mapped onto urf_libm.h (oracle/_ref/urf_ref_libm: what the product equals bit for bit), and counts the labels that differ.

    python tools/glibc_libm_diff.py [--sweeps 7680] [--procs 8]        (CPU only; needs /root/reference for the oracle-A builds)
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KINDS = (("sensor", 3, "cfg2"), ("sensor_narrow", 4, "cfg2"), ("sensor_default_roi", 3, "default_roi"), ("analytic", 1, "cfg2"))
CHUNK = 12


def work(task):
    import oracles as O
    import urban_road_filter_amd as u
    kind, seed0 = task
    name, scene, pname = KINDS[kind]
    p = O.cfg_params(pname)
    scans = [u.synth_cloud(64, 2048, scene, seed0 + k) for k in range(CHUNK)]
    la, _, _, _ = O.run_a(scans, p, libm=False)
    lb, _, _, _ = O.run_a(scans, p, libm=True)
    diff = sum(int(np.count_nonzero(a != b)) for a, b in zip(la, lb))
    sweeps_hit = sum(1 for a, b in zip(la, lb) if not np.array_equal(a, b))
    return kind, CHUNK * 64 * 2048, diff, sweeps_hit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=7680)
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    import oracles as O
    assert O.has_oracle_a() and os.path.exists(O.ORACLE_A_LIBM), "needs oracle/_ref (the reference's sources under /root/reference)"
    tasks = [(i % len(KINDS), 6_000_000 + i * CHUNK) for i in range((a.sweeps + CHUNK - 1) // CHUNK)]
    t0 = time.time()
    tot = {k: [0, 0, 0] for k in range(len(KINDS))}
    with mp.Pool(a.procs) as pool:
        for kind, pts, diff, hit in pool.imap_unordered(work, tasks):
            tot[kind][0] += pts
            tot[kind][1] += diff
            tot[kind][2] += hit
    print("# reference + glibc (%s) against reference + include/urf_libm.h, labels that differ" % os.confstr("CS_GNU_LIBC_VERSION"))
    allp = alld = 0
    for k, (name, scene, pname) in enumerate(KINDS):
        pts, diff, hit = tot[k]
        allp += pts
        alld += diff
        print("%-20s scene %d params %-12s %14d points  %6d labels differ  (%d sweeps affected)  = %.2f per 10^9 points" % (
            name, scene, pname, pts, diff, hit, 1e9 * diff / max(pts, 1)))
    print("total %d points, %d labels differ = %.2f per 10^9 points  (%.0f s on %d processes)" % (allp, alld, 1e9 * alld / max(allp, 1), time.time() - t0, a.procs))


if __name__ == "__main__":
    main()
