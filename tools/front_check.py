#!/usr/bin/env python3
"""The fused front end (urban_road_filter_amd/csrc/urf_front.hpp) against oracle B and against the legacy kernels, on the GPU.

    python tools/front_check.py [--fuzz N] [--seed0 S] [--basic 0|1]

Every case is run as a batch call with front mode 2 (fused wherever the scan has the shape) and with mode 0 (legacy kernels);
labels and summaries of both must equal oracle B's.  Prints how many scans of each case took the fused path."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import urban_road_filter_amd as u   # noqa: E402
from hipmem import DevBuf           # noqa: E402
import oracles                      # noqa: E402
import fuzz_organised               # noqa: E402

INFO_KEYS = ("status", "n_roi", "n_rings", "n_ring_pts", "n_road", "n_curb", "n_ring10", "n_nan_azimuth")


def run_batch(ctx, sweeps, mode):
    n = len(sweeps[0][0])
    S = len(sweeps)
    X = np.concatenate([s[0] for s in sweeps]).astype(np.float32)
    Y = np.concatenate([s[1] for s in sweeps]).astype(np.float32)
    Z = np.concatenate([s[2] for s in sweeps]).astype(np.float32)
    dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
    dl = DevBuf(n * S)
    dl.fill(0xEE)
    di = DevBuf(S * 32)
    ctx.set_front_mode(mode)
    ctx.classify_batch_soa(dx, dy, dz, n, S, dl, di)
    ctx.synchronize()
    nf = ctx.front_scans()
    lab = dl.to_numpy(np.uint8).reshape(S, n)
    info = di.to_numpy(np.uint32).reshape(S, 8)
    return lab, info, nf


def check(name, sweeps, params, ctxs, verbose=True, fresh=False):
    """fresh: a context of its own -- its first call with a scan the fused front end hands back goes through the LIST-driven
    legacy kernels (a context that has seen one launches them as full grids from then on)."""
    n = len(sweeps[0][0])
    key = (n, len(sweeps), name if fresh else "")
    if key not in ctxs:
        ctxs[key] = u.Context(n, len(sweeps))
    ctx = ctxs[key]
    ctx.set_params(params)
    bad = 0
    res = {}
    for mode in (2, 3, 0):   # 3 = mode 2 once more: a context that has handed a scan back launches the legacy kernels as full grids
        lab, info, nf = run_batch(ctx, sweeps, min(mode, 2))
        res[mode] = (lab, info, nf)
    for si, (x, y, z) in enumerate(sweeps):
        lb, ib, _ = oracles.run_b(x, y, z, params)
        for mode in (2, 3, 0):
            lab, info, nf = res[mode]
            d = int(np.count_nonzero(lab[si] != lb))
            iv = dict(zip(INFO_KEYS, [int(v) for v in info[si]]))
            iv["status"] = int(np.int32(info[si][0]))
            di = {k: (iv[k], ib[k]) for k in INFO_KEYS if iv[k] != ib[k]}
            if d or di:
                bad += 1
                if verbose:
                    w = np.nonzero(lab[si] != lb)[0]
                    print("  MISMATCH %s scan %d mode %d: %d labels differ %s first %s gpu %s ref %s" % (
                        name, si, mode, d, di, w[:8], lab[si][w[:8]], lb[w[:8]]))
    print("%-44s scans %3d fused %3d / %3d  %s" % (name, len(sweeps), res[2][2], res[3][2], "ok" if not bad else "FAILED (%d)" % bad), flush=True)
    return bad


def permute_lanes(sw, perm):
    out = []
    for a in sw:
        out.append(np.ascontiguousarray(a.reshape(-1, 64)[:, perm].reshape(-1)))
    return tuple(out)


def to_rows(sw):
    """firing order -> the row-major organised layout (height = 64 lasers, width = firings)"""
    return tuple(np.ascontiguousarray(a.reshape(-1, 64).T.reshape(-1)) for a in sw)


def rotate_cols(sw, k):
    return tuple(np.ascontiguousarray(np.roll(a.reshape(-1, 64), k, axis=0).reshape(-1)) for a in sw)


def basic(ctxs):
    bad = 0
    P = oracles.cfg_params
    bad += check("cfg2 x4 (scene 1, wide roi)", [oracles.cfg_cloud("cfg2", s) for s in (1, 2, 3, 4)], P("cfg2"), ctxs)
    bad += check("narrow x3 (scene 2)", [oracles.cfg_cloud("narrow", s) for s in (1, 2, 3)], P("narrow"), ctxs)
    bad += check("sensor x4 (scene 3: holes, ties)", [oracles.cfg_cloud("sensor", s) for s in (1, 2, 3, 4)], P("sensor"), ctxs)
    bad += check("sensor_narrow x2", [oracles.cfg_cloud("sensor_narrow", s) for s in (1, 2)], P("sensor_narrow"), ctxs)
    bad += check("default roi x3", [oracles.cfg_cloud("default_roi", s) for s in (1, 2, 3)], P("default_roi"), ctxs)
    bad += check("sensor default roi x2", [oracles.cfg_cloud("sensor_default_roi", s) for s in (1, 2)], P("sensor_default_roi"), ctxs)
    rng = np.random.default_rng(5)
    perm = rng.permutation(64)
    bad += check("laser order (lanes permuted) x3", [permute_lanes(oracles.cfg_cloud("sensor", s), perm) for s in (1, 2, 3)], P("sensor"), ctxs)
    bad += check("rotated start column x3", [rotate_cols(oracles.cfg_cloud("cfg2", s), k) for s, k in ((1, 5), (2, 700), (3, 1999))], P("cfg2"), ctxs, fresh=True)
    p = P("cfg2")
    for xd in (1, 2):
        p.xDirection = xd
        bad += check("narrow xDirection %d" % xd, [oracles.cfg_cloud("narrow", 1)], p, ctxs)
    p = P("cfg2")
    p.starbeam_filter = 1
    bad += check("starbeam filter", [oracles.cfg_cloud("cfg2", 1), oracles.cfg_cloud("sensor", 2)], p, ctxs)
    p = P("cfg2")
    p.star_shaped_method = 0
    bad += check("no star", [oracles.cfg_cloud("sensor", 1)], p, ctxs)
    p = P("cfg2")
    p.curbHeight = 0.01
    bad += check("rough: curbHeight 0.01 (lists overflow)", [oracles.cfg_cloud("sensor", 1), oracles.cfg_cloud("cfg2", 2)], p, ctxs)
    # mixed batch: a shuffled sweep next to organised ones
    sw = [oracles.cfg_cloud("cfg2", 1), oracles.cfg_cloud("sensor", 2)]
    x, y, z = oracles.cfg_cloud("cfg2", 3)
    pm = np.random.default_rng(1).permutation(len(x))
    sw.insert(1, (x[pm], y[pm], z[pm]))
    bad += check("mixed batch (scan 1 shuffled), lists", sw, P("cfg2"), ctxs, fresh=True)
    bad += check("mixed batch (scan 1 shuffled), grids", sw, P("cfg2"), ctxs)
    # short sweeps: 64 x 96 (one and a half tiles ... partial last tile), 64 x 40
    for cols in (96, 40, 33):
        bad += check("short sweep 64 x %d" % cols, [u.synth_cloud(64, cols, 1, 7)], P("cfg2"), ctxs, fresh=True)
    # the reference's default region of interest on a sweep stored from another column: the speculative ring table fails
    sw = [rotate_cols(oracles.cfg_cloud("default_roi", s), 1024) for s in (1, 2)] + [oracles.cfg_cloud("default_roi", 3)]
    bad += check("default roi, rear-stored (table repair), lists", sw, P("default_roi"), ctxs, fresh=True)
    bad += check("default roi, rear-stored (table repair), again", sw, P("default_roi"), ctxs, fresh=True)
    return bad


def rows(ctxs):
    """row-major organised sweeps: the first call of a context only sights them (legacy kernels), the second takes k_transpose + the fused kernels"""
    bad = 0
    P = oracles.cfg_params
    R = to_rows
    bad += check("rows: cfg2 x4", [R(oracles.cfg_cloud("cfg2", s)) for s in (1, 2, 3, 4)], P("cfg2"), ctxs, fresh=True)
    bad += check("rows: sensor x4 (holes, ties)", [R(oracles.cfg_cloud("sensor", s)) for s in (1, 2, 3, 4)], P("sensor"), ctxs)
    bad += check("rows: narrow x3", [R(oracles.cfg_cloud("narrow", s)) for s in (1, 2, 3)], P("narrow"), ctxs, fresh=True)
    bad += check("rows: sensor_narrow x2", [R(oracles.cfg_cloud("sensor_narrow", s)) for s in (1, 2)], P("sensor_narrow"), ctxs, fresh=True)
    bad += check("rows: default roi x3", [R(oracles.cfg_cloud("default_roi", s)) for s in (1, 2, 3)], P("default_roi"), ctxs, fresh=True)
    bad += check("rows: sensor default roi x2", [R(oracles.cfg_cloud("sensor_default_roi", s)) for s in (1, 2)], P("sensor_default_roi"), ctxs, fresh=True)
    sw = [R(rotate_cols(oracles.cfg_cloud("default_roi", s), 1024)) for s in (1, 2)] + [R(oracles.cfg_cloud("default_roi", 3))]
    bad += check("rows: default roi, rear-stored", sw, P("default_roi"), ctxs, fresh=True)
    perm = np.random.default_rng(5).permutation(64)
    bad += check("rows: lasers permuted x3", [R(permute_lanes(oracles.cfg_cloud("sensor", s), perm)) for s in (1, 2, 3)], P("sensor"), ctxs, fresh=True)
    p = P("cfg2")
    p.starbeam_filter = 1
    bad += check("rows: starbeam filter", [R(oracles.cfg_cloud("cfg2", 1)), R(oracles.cfg_cloud("sensor", 2))], p, ctxs, fresh=True)
    p = P("cfg2")
    p.curbHeight = 0.01
    bad += check("rows: rough (lists overflow)", [R(oracles.cfg_cloud("sensor", 1)), R(oracles.cfg_cloud("cfg2", 2))], p, ctxs, fresh=True)
    x, y, z = oracles.cfg_cloud("cfg2", 3)
    pm = np.random.default_rng(1).permutation(len(x))
    sw = [R(oracles.cfg_cloud("cfg2", 1)), (x[pm], y[pm], z[pm]), oracles.cfg_cloud("sensor", 2), R(oracles.cfg_cloud("sensor", 4))]
    bad += check("rows: mixed batch (rows, shuffled, firings, rows)", sw, P("cfg2"), ctxs, fresh=True)
    for cols in (96, 40, 33, 8):
        bad += check("rows: short sweep 64 x %d" % cols, [R(u.synth_cloud(64, cols, 1, 7))], P("cfg2"), ctxs, fresh=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzz", type=int, default=0)
    ap.add_argument("--seed0", type=int, default=9000000)
    ap.add_argument("--basic", type=int, default=1)
    ap.add_argument("--rows", type=int, default=1, help="the row-major cases; with --fuzz: 1 = every case is also run row-major")
    ap.add_argument("--callback", type=int, default=0, help="with --fuzz and --rows: every row-major case also as one sweep on the callback path")
    a = ap.parse_args()
    ctxs = {}
    bad = 0
    t0 = time.time()
    if a.basic:
        bad += basic(ctxs)
        if a.rows:
            bad += rows(ctxs)
    nfused = 0
    nrows = 0
    ncb = 0
    rctx = {}
    cctx = {}
    for i in range(a.fuzz):
        seed = a.seed0 + i
        sw, p = fuzz_organised.case(seed)
        n = len(sw[0])
        key = (n, 1)
        if key not in ctxs:
            ctxs[key] = u.Context(n, 1)
        ctx = ctxs[key]
        ctx.set_params(p)
        lb, ib, _ = oracles.run_b(sw[0], sw[1], sw[2], p)
        lab, info, nf = run_batch(ctx, [sw], 2)
        nfused += nf
        iv = dict(zip(INFO_KEYS, [int(v) for v in info[0]]))
        d = int(np.count_nonzero(lab[0] != lb))
        di = {k: (iv[k], ib[k]) for k in INFO_KEYS if iv[k] != ib[k]}
        if d or di:
            bad += 1
            print("  FUZZ MISMATCH seed %d fused %d: %d labels %s" % (seed, nf, d, di), flush=True)
        if a.rows and n % 64 == 0:
            # the same sweep row-major (a context of their own: its first call sights the layout, the following ones take it)
            rs = to_rows(sw)
            if key not in rctx:
                rctx[key] = u.Context(n, 1)
            rc = rctx[key]
            rc.set_params(p)
            lbr, ibr, _ = oracles.run_b(rs[0], rs[1], rs[2], p)
            lab, info, nf = run_batch(rc, [rs], 2)
            nrows += nf
            iv = dict(zip(INFO_KEYS, [int(v) for v in info[0]]))
            iv["status"] = int(np.int32(info[0][0]))
            d = int(np.count_nonzero(lab[0] != lbr))
            di = {k: (iv[k], ibr[k]) for k in INFO_KEYS if iv[k] != ibr[k]}
            if d or di:
                bad += 1
                print("  FUZZ MISMATCH (row-major) seed %d fused %d: %d labels %s" % (seed, nf, d, di), flush=True)
            if a.callback:
                # ... and as ONE sweep on the callback path (urf_classify_pc2: the fused kernels inside the captured sequence once the
                # context's sweeps have turned out row-major)
                if key not in cctx:
                    cctx[key] = u.Context(n, 4)
                cc = cctx[key]
                cc.set_params(p)
                lab1, info1 = cc.classify_xyz(*rs)
                ncb += cc.front_scans()
                iv = {k: int(getattr(info1, k)) for k in INFO_KEYS}
                d = int(np.count_nonzero(lab1 != lbr))
                di = {k: (iv[k], ibr[k]) for k in INFO_KEYS if iv[k] != ibr[k]}
                if d or di:
                    bad += 1
                    print("  FUZZ MISMATCH (row-major, callback path) seed %d: %d labels %s" % (seed, d, di), flush=True)
    if a.fuzz:
        print("fuzz: %d cases, %d took the fused front end, %d of their row-major twins, %d mismatches, %.0f s" % (a.fuzz, nfused, nrows, bad, time.time() - t0))
        if a.callback:
            print("      %d of the row-major twins took it on the callback path" % ncb)
    print("FRONT CHECK %s" % ("PASSED" if bad == 0 else "FAILED: %d" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
