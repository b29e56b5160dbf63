#!/usr/bin/env python3
"""Batches of 4 .. 256 resident sweeps through the general kernels (front mode 0) and the fused front end (mode 2): where mode 1's threshold
(URF_FRONT_MIN_SCANS) belongs.  python tools/r6_min_scans.py [--rows]"""
import os, sys, time, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import urban_road_filter_amd as u, oracles as O
from hipmem import DevBuf
n = 64 * 2048
p = O.cfg_params("cfg2")
ROWS = "--rows" in sys.argv   # the same sweeps row-major (height = 64)
for S in (4, 8, 16, 32, 64, 128, 256):
    cl = [u.synth_cloud(64, 2048, 1, 1 + s) for s in range(S)]
    if ROWS:
        cl = [tuple(np.ascontiguousarray(a.reshape(-1, 64).T.reshape(-1)) for a in c) for c in cl]
    X, Y, Z = (np.concatenate(a) for a in zip(*cl))
    dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z); dl = DevBuf(S * n)
    out = []
    for mode in (0, 2):
        ctx = u.Context(n, S, params=p); ctx.set_front_mode(mode)
        for _ in range(5): ctx.classify_batch_soa(dx, dy, dz, n, S, dl, None)
        ctx.synchronize()
        reps = max(20, 2000 // S)
        t0 = time.perf_counter()
        for _ in range(reps): ctx.classify_batch_soa(dx, dy, dz, n, S, dl, None)
        ctx.synchronize(); out.append((time.perf_counter() - t0) * 1e3 / reps)
        ctx.close()
    print("%4d sweeps per call: general %.4f ms, fused %.4f ms (%+.0f %%)" % (S, out[0], out[1], 100 * (out[1] / out[0] - 1)), flush=True)
    for b in (dx, dy, dz, dl): b.free()
