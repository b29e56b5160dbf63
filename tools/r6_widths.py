#!/usr/bin/env python3
"""Sweeps of 64 x 512 / 1024 / 2048 / 4096 points in batches of 256 and 1024 through the general (front mode 0) and the fused kernels (mode 2), firing
order and row-major.  python tools/r6_widths.py"""
import os, sys, time, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import urban_road_filter_amd as u, oracles as O
from hipmem import DevBuf
p = O.cfg_params("cfg2")
for W in (512, 1024, 2048, 4096):
    n = 64 * W
    for S in (256, 1024):
        if n * S > 64 * 2048 * 1024:
            continue
        base = [u.synth_cloud(64, W, 1, 1 + s) for s in range(16)]
        for layout in ("firing", "rows"):
            cl = base if layout == "firing" else [tuple(np.ascontiguousarray(a.reshape(-1, 64).T.reshape(-1)) for a in c) for c in base]
            X, Y, Z = (np.concatenate([cl[s % 16][k] for s in range(S)]) for k in range(3))
            dx, dy, dz = DevBuf.from_numpy(X), DevBuf.from_numpy(Y), DevBuf.from_numpy(Z); dl = DevBuf(S * n)
            out = []
            for mode in (0, 2):
                ctx = u.Context(n, S, params=p); ctx.set_front_mode(mode)
                for _ in range(4): ctx.classify_batch_soa(dx, dy, dz, n, S, dl, None)
                ctx.synchronize()
                nf = ctx.front_scans()
                t0 = time.perf_counter()
                for _ in range(10): ctx.classify_batch_soa(dx, dy, dz, n, S, dl, None)
                ctx.synchronize(); out.append((time.perf_counter() - t0) * 100)
                ctx.close()
            print("64 x %4d, %4d sweeps, %-6s: general %.3f ms, fused %.3f ms (%+.0f %%), fused scans %d" % (W, S, layout, out[0], out[1], 100 * (out[1] / out[0] - 1), nf), flush=True)
            for b in (dx, dy, dz, dl): b.free()
