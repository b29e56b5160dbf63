#!/usr/bin/env python3
"""Sweeps stored ring by ring (row-major 64 x W) against oracle B, and timed: python tools/r6_ring_major.py [--scans 256]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import urban_road_filter_amd as u
from hipmem import DevBuf
import oracles

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=64)
ap.add_argument("--check", type=int, default=4)
a = ap.parse_args()
S, n = a.scans, 64 * 2048
idx = np.arange(n).reshape(2048, 64).T.reshape(-1)
clouds = []
for s in range(S):
    name = ("cfg2", "sensor", "narrow", "default_roi")[s % 4] if s < 8 else "cfg2"
    x, y, z = oracles.cfg_cloud(name, 1 + s)
    clouds.append((x[idx].copy(), y[idx].copy(), z[idx].copy()))
for pname in ("cfg2", "default_roi"):
    p = oracles.cfg_params(pname)
    ctx = u.Context(n, S, params=p)
    X, Y, Z = (np.concatenate([c[k] for c in clouds]) for k in range(3))
    dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z)
    dl, di = DevBuf(S * n), DevBuf(S * 32)
    ctx.classify_batch_soa(dx, dy, dz, n, S, dl, di)
    ctx.synchronize()
    L = dl.to_numpy(np.uint8).reshape(S, n)
    bad = 0
    for s in range(min(a.check, S)):
        lb, ib, _ = oracles.run_b(*clouds[s], p)
        d = int((L[s] != lb).sum())
        bad += d > 0
        if d:
            print("  scan %d (%s): %d labels differ" % (s, pname, d))
    ctx.enable_kernel_timing(True); ctx.kernel_timing(); ctx.enable_kernel_timing(True)
    t0 = time.time()
    for _ in range(5):
        ctx.classify_batch_soa(dx, dy, dz, n, S, dl, di)
    ctx.synchronize()
    kms, kc = ctx.kernel_timing()
    print("%s: %d scans, parity %s, %.3f ms per step  " % (pname, S, "ok" if not bad else "FAILED", sum(kms.values()) / kc),
          " ".join("%s=%.3f" % (k[2:], v / kc) for k, v in kms.items()), flush=True)
    ctx.close()
