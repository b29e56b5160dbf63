"""r5 experiment (review item 2 i): does the step get faster when it runs as SUB-BATCHES that reuse the same scratch rows, so
that what k_split writes (26 B/pt) and its three readers touch the 256 MB Infinity Cache instead of HBM?
One context per configuration with max_batch = sub, all sub-batches on one stream (or two contexts on two streams,
alternating: the next sub-batch's first kernels overlap the tail of the one before)."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import torch

import bench
import oracles as O
import urban_road_filter_amd as u

S = 1024
N = bench.N_PTS
X, Y, Z = bench.gen_batch(S, 1)
dev = torch.device("cuda", 0)
dx, dy, dz = [torch.from_numpy(a).to(dev) for a in (X, Y, Z)]
dl = torch.empty((S, N), dtype=torch.uint8, device=dev)
p = O.cfg_params("cfg2")
ref = None
for sub, nctx in ((1024, 1), (256, 1), (128, 1), (64, 1), (32, 1), (16, 1), (128, 2), (64, 2), (32, 2), (64, 4)):
    ctxs = [u.Context(N, sub, device=0, params=p) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    for c, st in zip(ctxs, streams):
        c.set_stream(st.cuda_stream)

    def step():
        for k in range(S // sub):
            lo = k * sub
            ctxs[k % nctx].classify_batch_soa(dx[lo:lo + sub], dy[lo:lo + sub], dz[lo:lo + sub], N, sub, dl[lo:lo + sub], None)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    L = dl.cpu().numpy()
    if ref is None:
        ref = L.copy()
    assert np.array_equal(L, ref)
    t = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print("sub-batch %4d x %d stream(s): %.3f ms/step  %.0f scans/s" % (sub, nctx, el / 10 * 1e3, S * 10 / el), flush=True)
    for c in ctxs:
        c.close()
