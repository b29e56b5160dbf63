#!/usr/bin/env python3
"""The headline's batch as ONE call of 1024 sweeps against TWO contexts of 512 sweeps each on two streams (does the vector-bound
k_front of one half run next to the latency-bound star search of the other?).  python tools/r6_two_streams.py [--scans 1024]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import urban_road_filter_amd as u
import oracles as O
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=1024)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--scene", type=int, default=1)
a = ap.parse_args()
S, N = a.scans, 64 * 2048
dev = torch.device("cuda:0")
X, Y, Z = bench.gen_batch(S, 1, 1, a.scene)
dx, dy, dz = (torch.from_numpy(t).to(dev) for t in (X, Y, Z))
dl = torch.empty((S, N), dtype=torch.uint8, device=dev)
p = O.cfg_params("cfg2")
torch.cuda.synchronize()


def timed(fn):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.steps


with u.Context(N, S, device=0, params=p) as c1:
    st = torch.cuda.Stream(device=dev)
    c1.set_stream(st.cuda_stream)
    one = timed(lambda: c1.classify_batch_soa(dx, dy, dz, N, S, dl, None))
    ref = dl.clone()
print("one context, %d sweeps per call:            %.4f ms per %d sweeps" % (S, one, S))
for parts in (2, 4):
    H = S // parts
    ctxs, streams = [], []
    for k in range(parts):
        c = u.Context(N, H, device=0, params=p)
        s = torch.cuda.Stream(device=dev)
        c.set_stream(s.cuda_stream)
        c.set_front_mode(2)
        ctxs.append(c); streams.append(s)

    def step():
        for k, c in enumerate(ctxs):
            c.classify_batch_soa(dx[k * H:(k + 1) * H], dy[k * H:(k + 1) * H], dz[k * H:(k + 1) * H], N, H, dl[k * H:(k + 1) * H], None)
    dl.zero_()
    torch.cuda.synchronize()
    ms = timed(step)
    same = bool(torch.equal(dl, ref))
    print("%d contexts of %d sweeps on %d streams:      %.4f ms per %d sweeps, labels equal: %s" % (parts, H, parts, ms, S, same))
    for c in ctxs:
        c.close()
