"""Multi-GPU partitioning of the hot path.

Scans are independent (the reference keeps no cross-scan state on this path), so a batch is
sharded by scan: every rank owns a contiguous block of scans, runs the full pipeline on its own
GPU and never exchanges point data.  The only communication is bookkeeping at the end of a run:
one all-reduce(SUM) over a handful of 64-bit counters and one all-reduce(MAX) of the elapsed time
(RCCL over xGMI on GPUs -- latency-bound, 48 bytes; gloo in the CPU tests).  A scan is never
split across GPUs: rings and sectors of one scan are coupled by the beam march and the ring table.
"""
import numpy as np


def shard_range(n_scans_total, rank, world):
    """Contiguous block [lo, hi) of the scans 0..n_scans_total-1 owned by `rank`; blocks differ in
    size by at most one scan."""
    base, rem = divmod(n_scans_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(scans_per_gpu, rank):
    """Weak-scaling benchmark: rank r generates the sweeps with seeds r*S+1 .. (r+1)*S."""
    return range(1 + rank * scans_per_gpu, 1 + (rank + 1) * scans_per_gpu)


COUNTER_NAMES = ("scans", "points_in", "roi_points", "road", "curb", "ok_scans")


def local_counters(info, n_points_per_scan, steps=1):
    """Totals over the whole run, every entry in the same unit (summed over all `steps` passes over
    the batch).  info: int array [S, 8] of urf_scan_info rows of ONE step; the batch is the same in
    every step, so each per-step sum counts `steps` times."""
    info = np.asarray(info, dtype=np.int64)
    s = info.shape[0]
    return steps * np.array([s, s * n_points_per_scan, info[:, 1].sum(), info[:, 4].sum(), info[:, 5].sum(),
                             (info[:, 0] == 0).sum()], dtype=np.int64)


def reduce_run(counters, elapsed_s, device=None):
    """all-reduce(SUM) of the counters and all-reduce(MAX) of the elapsed time over the default
    process group; identity when torch.distributed is not initialised.  Returns (counters, t_max)."""
    import torch
    import torch.distributed as dist
    c = torch.as_tensor(np.asarray(counters, dtype=np.int64), device=device)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():   # (also with one rank: bench.py --force-dist runs the collectives on one GPU)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return c.cpu().numpy(), float(t.item())
