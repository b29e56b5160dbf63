"""The N > 1 path on CPU: two processes, gloo backend.  Covers the scan partitioning and the
end-of-run counter / timing reduction that bench.py performs over RCCL on GPUs."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from urban_road_filter_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 8, 1024, 8192, 8193):
        for world in (1, 2, 3, 8):
            blocks = [sharding.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    seeds = [set(sharding.shard_seeds(1024, r)) for r in range(8)]
    assert set.union(*seeds) == set(range(1, 8193)) and sum(len(s) for s in seeds) == 8192


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import torch.distributed as dist
    import oracles as O
    import urban_road_filter_amd as u
    from urban_road_filter_amd import sharding
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 5
    lo, hi = sharding.shard_range(total, rank, world)
    p = O.cfg_params("cfg1")
    rows = []
    for s in range(lo, hi):            # each rank classifies ITS scans (CPU oracle stands in for the GPU here)
        x, y, z = O.cfg_cloud("cfg1", 1 + s)
        _, ib, _ = O.run_b(x, y, z, p)
        rows.append([ib["status"], ib["n_roi"], ib["n_rings"], ib["n_ring_pts"], ib["n_road"], ib["n_curb"], ib["n_ring10"], 0])
    c = sharding.local_counters(np.array(rows).reshape(-1, 8), 16 * 1024)
    c, tmax = sharding.reduce_run(c, 0.5 + rank)
    if rank == 0:
        print("RESULT " + json.dumps({"counters": c.tolist(), "tmax": tmax, "world": world}))
    dist.destroy_process_group()
""")


def test_two_rank_reduction_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0]
    import json
    res = json.loads(line[7:])
    assert res["world"] == 2 and res["tmax"] == 1.5
    # single-process ground truth over all 5 scans
    import oracles as O
    p = O.cfg_params("cfg1")
    road = roi = 0
    for s in range(5):
        _, ib, _ = O.run_b(*O.cfg_cloud("cfg1", 1 + s), p)
        road += ib["n_road"]
        roi += ib["n_roi"]
    scans, pts, roi_g, road_g, curb_g, ok_g = res["counters"]
    assert (scans, pts, roi_g, road_g, ok_g) == (5, 5 * 16384, roi, road, 5)
