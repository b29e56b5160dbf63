#!/usr/bin/env python3
"""Fused front end against the legacy kernels on a whole benchmark batch: which scans differ, and where.
    python tools/front_diff.py [--scans 1024] [--scene 1] [--params cfg2] [--reps 3]"""
import argparse
import concurrent.futures as cf
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import urban_road_filter_amd as u   # noqa: E402
from hipmem import DevBuf           # noqa: E402
import oracles                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=1024)
    ap.add_argument("--scene", type=int, default=1)
    ap.add_argument("--params", default="cfg2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--perm", type=int, default=-1, help="seed of a fixed permutation of the 64 lasers inside every firing")
    ap.add_argument("--oracle", default="", help="comma-separated scans to compare with oracle B as well")
    ap.add_argument("--fresh", type=int, default=0, help="this many fresh contexts whose FIRST call takes the fused front end")
    a = ap.parse_args()
    S, n = a.scans, 64 * 2048
    X = np.empty((S, n), np.float32)
    Y = np.empty_like(X)
    Z = np.empty_like(X)

    def one(s):
        X[s], Y[s], Z[s] = u.synth_cloud(64, 2048, a.scene, 1 + s)

    with cf.ThreadPoolExecutor(max_workers=32) as ex:
        list(ex.map(one, range(S)))
    if a.perm >= 0:
        pm = np.random.default_rng(a.perm).permutation(64)
        X, Y, Z = (np.ascontiguousarray(v.reshape(S, 2048, 64)[:, :, pm].reshape(S, n)) for v in (X, Y, Z))
    p = oracles.cfg_params(a.params)
    ctx = u.Context(n, S, params=p)
    dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
    dl, di = DevBuf(S * n), DevBuf(S * 32)
    ctx.set_front_mode(0)
    ctx.classify_batch_soa(dx, dy, dz, n, S, dl, di)
    ctx.synchronize()
    L0 = dl.to_numpy(np.uint8).reshape(S, n).copy()
    I0 = di.to_numpy(np.uint32).reshape(S, 8).copy()
    total_bad = 0
    for sc in [int(v) for v in a.oracle.split(",") if v]:
        lb, ib, _ = oracles.run_b(X[sc], Y[sc], Z[sc], p)
        d = np.nonzero(L0[sc] != lb)[0]
        print("oracle vs legacy scan %d: %d labels differ %s legacy %s oracle %s" % (sc, len(d), d[:10], L0[sc][d[:10]], lb[d[:10]]), flush=True)
        total_bad += len(d) > 0
    for rep in range(a.reps):
        dl.fill(0xEE)
        ctx.set_front_mode(2)
        ctx.classify_batch_soa(dx, dy, dz, n, S, dl, di)
        ctx.synchronize()
        nf = ctx.front_scans()
        L1 = dl.to_numpy(np.uint8).reshape(S, n)
        I1 = di.to_numpy(np.uint32).reshape(S, 8)
        bad = [s for s in range(S) if not np.array_equal(L0[s], L1[s]) or not np.array_equal(I0[s], I1[s])]
        total_bad += len(bad)
        print("rep %d: fused %d of %d, %d scans differ" % (rep, nf, S, len(bad)), flush=True)
        for s in bad[:6]:
            w = np.nonzero(L0[s] != L1[s])[0]
            print("  scan %d: %d labels; info legacy %s fused %s" % (s, len(w), I0[s].tolist(), I1[s].tolist()))
            if len(w):
                rings, fir = w % 64, w // 64
                print("    lanes %s firings %s..%s  legacy %s fused %s" % (sorted(set(rings.tolist()))[:20], fir.min(), fir.max(),
                                                                       L0[s][w[:12]].tolist(), L1[s][w[:12]].tolist()))
                print("    firings", sorted(set(fir.tolist()))[:40])
    for k in range(a.fresh):
        c2 = u.Context(n, S, params=p)
        dl.fill(0x55)
        c2.set_front_mode(2)
        c2.classify_batch_soa(dx, dy, dz, n, S, dl, di)
        c2.synchronize()
        nf = c2.front_scans()
        L1 = dl.to_numpy(np.uint8).reshape(S, n)
        I1 = di.to_numpy(np.uint32).reshape(S, 8)
        bad = [s for s in range(S) if not np.array_equal(L0[s], L1[s]) or not np.array_equal(I0[s], I1[s])]
        total_bad += len(bad)
        print("fresh %d: fused %d of %d, %d scans differ" % (k, nf, S, len(bad)), flush=True)
        for s in bad[:6]:
            w = np.nonzero(L0[s] != L1[s])[0]
            print("  scan %d: %d labels; info legacy %s fused %s" % (s, len(w), I0[s].tolist(), I1[s].tolist()))
            if len(w):
                rings, fir = w % 64, w // 64
                print("    lanes %s firings %s..%s  legacy %s fused %s" % (sorted(set(rings.tolist()))[:20], fir.min(), fir.max(),
                                                                       L0[s][w[:12]].tolist(), L1[s][w[:12]].tolist()))
                print("    firings", sorted(set(fir.tolist()))[:40])
        c2.close()
    print("FRONT DIFF %s" % ("PASSED" if total_bad == 0 else "FAILED"))
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
