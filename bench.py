#!/usr/bin/env python3
"""bench.py -- scans/sec of the urban_road_filter classification hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scans S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one pass of the whole hot path (all kernels of urf_classify_batch_soa: ROI, ring
table, ring split, star-shaped search, x_zero, z_zero, blind-spot beam march, label write-back)
over one HBM-resident batch of S = 1024 independent synthetic 64-ring x 2048-column sweeps
(BASELINE.json configs[2] "throughput saturation", the configuration the scans/sec metric is
quoted on).  With N GPUs every rank owns its own batch of S sweeps (weak
scaling, no collective on the data path; RCCL only reduces timing and counters -- SURVEY.md 8e).

Prints ONE JSON line (rank 0).  value = N*S*K / t, t = max over ranks of the wall time of exactly
K steps bracketed by barrier + device synchronize.  Inputs are resident in HBM before the timed
region; the PCIe-inclusive rate is reported separately as `h2d_inclusive_scans_per_s`, and the
callback path (one sweep: host PointCloud2 bytes -> host labels) as `e2e_*`.

--backend gloo (or URF_BENCH_BACKEND=gloo) runs the N-rank launch path without RCCL and lets ranks
share a device (rank r uses device r mod #devices): `--gpus 2 --backend gloo` exercises
torch.distributed.run, one context per rank, disjoint seeds and the counter reduction on a 1-GPU box.
The default backend is nccl (= RCCL on ROCm).
"""
import argparse
import concurrent.futures as cf
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RINGS, COLS = 64, 2048
N_PTS = RINGS * COLS
ALG_BYTES_PER_SCAN = 13 * N_PTS          # 12 B x,y,z read + 1 B label written per point
HBM_PEAK_GBS = 8000.0                    # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# The default workload is cfg3 (the one the metric is quoted on).  The others are the remaining
# BASELINE.json configurations, for the per-configuration table in profiles/ (not bench lines).
WORKLOADS = {
    "cfg3": dict(rings=64, cols=2048, scans=1024, params="cfg2",
                 text="cfg3: batch of %d independent 64x2048 street sweeps per GPU, all three detectors + blind_spots, "
                      "reference default parameters with ROI x,y widened to +-200 m, inputs resident in HBM (SoA x/y/z)"),
    "cfg2": dict(rings=64, cols=2048, scans=1, params="cfg2",
                 text="cfg2: %d single 64x2048 street sweep, all three detectors + blind_spots, ROI +-200 m, resident in HBM (latency)"),
    "cfg5": dict(rings=128, cols=4096, scans=256, params="cfg5",
                 text="cfg5: batch of %d 128x4096 street sweeps, channels=128, interval=0.05, star at 360 sectors, ROI +-200 m"),
    "default_roi": dict(rings=64, cols=2048, scans=1024, params="default_roi",
                        text="batch of %d 64x2048 street sweeps with the reference's DEFAULT ROI (x 0..30, y -10..10: ~40 %% of the points survive)"),
}


def gen_batch(n_scans, seed0):
    """n_scans distinct street sweeps (seeds seed0..), generated on a thread pool
    (urf_synth_cloud releases the GIL)."""
    import urban_road_filter_amd as u
    X = np.empty((n_scans, N_PTS), np.float32)
    Y = np.empty_like(X)
    Z = np.empty_like(X)

    def one(s):
        x, y, z = u.synth_cloud(RINGS, COLS, 1, seed0 + s)
        X[s], Y[s], Z[s] = x, y, z

    with cf.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(one, range(n_scans)))
    return X, Y, Z


CPU_BASELINE_MAX_PROCS = 64


def cpu_baseline(params, budget_scans=6):
    """The reference's own CPU path (oracle/_ref, built from its unmodified sources) timed on the
    host cores of this box: single process, and P = min(cores, 64) independent processes (the
    reference is single-threaded and keeps its state in globals).  A bounded sample: each process
    classifies `budget_scans` sweeps after one excluded warm-up call (first-touch of its 512 MiB
    scratch).  P is capped at 64 because every process value-initialises a 512 MiB matrix per sweep
    (lidar_segmentation.cpp:207): beyond a few dozen processes the page-fault path of the kernel, not
    the cores, sets the rate, and 256 x 512 MiB would not be a bounded sample any more.
    Falls back to the C restatement (kind "port") when the binary is absent."""
    import oracles as O
    cores_avail = os.cpu_count() or 1
    scans = [O.cfg_cloud("cfg2", 1 + s) for s in range(2)]
    if O.has_oracle_a():
        kind = "reference"
        _, _, ms1, _ = O.run_a(scans, params, repeat=1 + budget_scans // 2)
        single = 1000.0 / ms1
        P = min(cores_avail, CPU_BASELINE_MAX_PROCS)
        # P concurrent processes
        with tempfile.TemporaryDirectory() as td:
            import struct
            fin = os.path.join(td, "in.bin")
            with open(fin, "wb") as f:
                f.write(b"URFREFIN")
                f.write(struct.pack("<4I", len(scans), N_PTS, 1 + budget_scans // 2, 0))
                f.write(bytes(params))
                for x, y, z in scans:
                    f.write(x.tobytes()); f.write(y.tobytes()); f.write(z.tobytes())
            procs = [subprocess.Popen([O.ORACLE_A, fin, os.path.join(td, "o%d.bin" % i)]) for i in range(P)]
            for pr in procs:
                pr.wait()
            rates = []
            for i in range(P):
                blob = open(os.path.join(td, "o%d.bin" % i), "rb").read()
                rates.append(1000.0 / struct.unpack_from("<d", blob, 16)[0])
        multi = float(sum(rates))
        sample = ("%d x 64x2048 street sweeps per process after 1 excluded warm-up call; "
                  "value = sum over %d concurrent single-threaded processes" % (2 * (1 + budget_scans // 2) - 1, P))
        return {"value": round(multi, 3), "unit": "scans/s", "cores": P, "kind": kind, "sample": sample,
                "single_core_value": round(single, 3), "host_cores_available": cores_avail,
                "cores_cap": "min(host cores, %d): each process value-initialises 512 MiB per sweep "
                             "(lidar_segmentation.cpp:207); more processes measure the kernel's page-fault path" % CPU_BASELINE_MAX_PROCS}
    kind = "port"
    t0 = time.perf_counter()
    k = 0
    while k < 3 * budget_scans:
        O.run_b(*scans[k % 2], params)
        k += 1
    single = k / (time.perf_counter() - t0)
    return {"value": round(single, 3), "unit": "scans/s", "cores": 1, "kind": kind,
            "sample": "%d x 64x2048 street sweeps, oracle/urf_oracle.c, one thread" % k,
            "host_cores_available": cores_avail}


def e2e_callback_path(u, O, params, n_sweeps=8, reps=40, stream_reps=160):
    """The reference's unit of work (lidar_segmentation.cpp:95-100, 612-621): ONE sweep arrives as
    PointCloud2 bytes in host memory (pcl::PointXYZI records, 32 bytes: 4 MiB per 64x2048 sweep), the
    labels return to host memory (128 KiB).  Latency of the synchronous entry point, and throughput
    with two sweeps in flight (urf_classify_pc2_async: copy of sweep i+1 overlaps kernels of sweep i)."""
    n = RINGS * COLS
    recs = []
    for k in range(n_sweeps):
        x, y, z = u.synth_cloud(RINGS, COLS, 1, 9000 + k)
        buf = np.zeros((n, 32), np.uint8)
        buf[:, 0:4] = x.view(np.uint8).reshape(-1, 4)
        buf[:, 4:8] = y.view(np.uint8).reshape(-1, 4)
        buf[:, 8:12] = z.view(np.uint8).reshape(-1, 4)
        recs.append(buf.reshape(-1))
    out = {"bytes_in_per_scan": int(recs[0].nbytes), "bytes_out_per_scan": n,
           "pcie_bound_scans_per_s": round(63e9 / (recs[0].nbytes + n), 1)}
    with u.Context(n, 2, params=params) as ctx:   # two scratch rows: the kernels of two sweeps overlap
        lab = np.empty(n, np.uint8)
        lb, _, _ = O.run_b(*u.synth_cloud(RINGS, COLS, 1, 9000), params)
        lg, _ = ctx.classify_pc2(recs[0], n, 32, 0, 4, 8)
        if not np.array_equal(lg, lb):
            raise SystemExit("parity failure on the callback path")

        def latency(flags):
            ctx.set_debug_flags(flags)
            for k in range(4):
                ctx.classify_pc2(recs[k % n_sweeps], n, 32, 0, 4, 8)
            ts = []
            for k in range(reps):
                t0 = time.perf_counter()
                t = ctx.classify_pc2_async(recs[k % n_sweeps], n, 32, 0, 4, 8)
                ctx.classify_pc2_wait(t, lab)
                ts.append(time.perf_counter() - t0)
            ctx.set_debug_flags(0)
            return 1e3 * float(np.median(ts))

        out["e2e_latency_ms"] = round(latency(0), 4)                  # graph replay
        out["e2e_latency_ms_kernel_by_kernel"] = round(latency(8), 4)  # the same without the captured graph

        def stream(zero_copy):
            inflight = []
            t0 = time.perf_counter()
            for k in range(stream_reps):
                if len(inflight) == 2:
                    ctx.classify_pc2_wait(inflight.pop(0), lab)
                rec = recs[k % n_sweeps]
                if zero_copy:   # the producer (a driver, a deserialiser) fills the pinned buffer itself
                    pin = ctx.pinned_input(rec.nbytes)
                    if k < 2:   # (the bench writes each of the two buffers once: producing the data is not what is timed)
                        pin[:] = rec
                    inflight.append(ctx.classify_pc2_async(pin.ctypes.data, n, 32, 0, 4, 8))
                else:
                    inflight.append(ctx.classify_pc2_async(rec, n, 32, 0, 4, 8))
            for t in inflight:
                ctx.classify_pc2_wait(t, lab)
            return stream_reps / (time.perf_counter() - t0)

        stream(False)
        out["e2e_overlapped_scans_per_s"] = round(stream(False), 1)
        out["e2e_overlapped_scans_per_s_pinned_producer"] = round(stream(True), 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scans", type=int, default=0, help="sweeps per GPU per step (default: the workload's, cfg3: 1024)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-scans", type=int, default=4)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=os.environ.get("URF_BENCH_BACKEND", "nccl"))
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-outputs", action="store_true")
    args = ap.parse_args()

    import torch   # first: the HIP runtime torch bundles is the one the C-ABI library binds to
    import torch.distributed as dist
    import oracles as O
    import urban_road_filter_amd as u
    from urban_road_filter_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch ourselves under torch.distributed.run
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the classification has no CPU path")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    global RINGS, COLS, N_PTS, ALG_BYTES_PER_SCAN
    wl = WORKLOADS[args.workload]
    RINGS, COLS = wl["rings"], wl["cols"]
    N_PTS = RINGS * COLS
    ALG_BYTES_PER_SCAN = 13 * N_PTS
    S = args.scans or wl["scans"]
    params = O.cfg_params(wl["params"])   # cfg3: reference defaults, ROI widened to +-200 m
    t_gen = time.perf_counter()
    X, Y, Z = gen_batch(S, sharding.shard_seeds(S, rank)[0])   # seeds 1..S on rank 0, S+1..2S on rank 1, ...
    t_gen = time.perf_counter() - t_gen

    stream = torch.cuda.Stream(device=dev)   # the library launches on this stream (urf_set_stream), so events on it see the kernels
    t_h2d = time.perf_counter()
    dx = torch.from_numpy(X).to(dev)
    dy = torch.from_numpy(Y).to(dev)
    dz = torch.from_numpy(Z).to(dev)
    torch.cuda.synchronize()
    t_h2d = time.perf_counter() - t_h2d
    dl = torch.empty((S, N_PTS), dtype=torch.uint8, device=dev)
    di = torch.zeros((S, 8), dtype=torch.int32, device=dev)

    ctx = u.Context(N_PTS, S, device=dev_index, params=params)
    ctx.set_stream(stream.cuda_stream)

    def step():
        ctx.classify_batch_soa(dx, dy, dz, N_PTS, S, dl, di)

    # parity gate: no number is reported for a batch whose labels differ from the CPU oracle
    step()
    torch.cuda.synchronize()
    picked = sorted(np.random.default_rng(20260925 + rank).choice(S, min(args.parity_scans, S), replace=False).tolist())
    for s in picked:   # a seeded random sample of the batch, not its first scans
        lb, _, _ = O.run_b(X[s], Y[s], Z[s], params)
        L = dl[s].cpu().numpy()
        if not np.array_equal(L, lb):
            raise SystemExit("parity failure on scan %d of rank %d: %d labels differ" % (s, rank, int((L != lb).sum())))

    for _ in range(args.warmup):
        step()
    ctx.enable_kernel_timing(True)
    ctx.kernel_timing()                     # reset
    ctx.enable_kernel_timing(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every step is also bracketed by events on the launch stream: their median is reported next to
    # the wall-clock mean that `value` is computed from
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(args.steps):
        step()
        ev[k + 1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    step_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    kms, kcalls = ctx.kernel_timing()
    ctx.enable_kernel_timing(False)

    # the optional outputs of SURVEY.md 8f for the whole batch (not part of `value`): index sets,
    # published order, marker points
    outputs_ms = None
    if world == 1 and args.workload == "cfg3" and not args.no_outputs:
        idx = [torch.empty((S, N_PTS), dtype=torch.int32, device=dev) for _ in range(4)]
        cnt = torch.zeros((S, 4), dtype=torch.int32, device=dev)
        mpts = torch.empty((S, 361, 4), dtype=torch.float32, device=dev)
        outputs_ms = {}
        with torch.cuda.stream(stream):
            for name, fn in (("compact_indices", lambda: ctx.compact_indices_batch(dl, N_PTS, S, idx[0], idx[1], idx[2], idx[3], cnt)),
                             ("ordered_indices", lambda: ctx.ordered_indices_batch(idx[0], idx[1], idx[2], N_PTS, cnt)),
                             ("marker_points", lambda: ctx.marker_points_batch(mpts, cnt))):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                fn()
                e1.record(stream)
                torch.cuda.synchronize()
                outputs_ms[name] = round(e0.elapsed_time(e1), 4)
        del idx, mpts

    # bookkeeping only: all-reduce(SUM) of six 64-bit counters + all-reduce(MAX) of the elapsed time
    # over RCCL; no point data ever crosses xGMI
    counters = sharding.local_counters(di.cpu().numpy(), N_PTS, steps=args.steps)
    counters, elapsed_max = sharding.reduce_run(counters, elapsed, device=dev if args.backend == "nccl" else None)

    if rank == 0:
        total_scans = int(counters[0])
        value = total_scans / elapsed_max
        ms_step = 1e3 * elapsed_max / args.steps
        dom = max(kms, key=lambda k: kms[k])
        dom_ms = kms[dom] / max(kcalls, 1)
        alg_bytes_launch = ALG_BYTES_PER_SCAN * S
        achieved = alg_bytes_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        whole = ALG_BYTES_PER_SCAN * S / (ms_step * 1e-3) / 1e9
        out = {
            "metric": "scans/sec (%d-ring x %d-column cloud)" % (RINGS, COLS),
            "value": round(value, 2),
            "unit": "scans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "ms_per_step_median_hipevent": round(float(np.median(step_ms)), 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32+f64",
            "data": "synthetic",
            "config": {"workload": wl["text"] % S,
                       "scans_per_gpu": S, "points_per_scan": N_PTS, "sharding": "one batch per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": dom, "kernel_ms": round(dom_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes_launch,
                         "whole_pipeline_achieved": round(whole, 2), "whole_pipeline_frac": round(whole / HBM_PEAK_GBS, 5)},
            "kernel_ms": {k: round(v / max(kcalls, 1), 4) for k, v in kms.items()},
            "outputs_ms_per_batch": outputs_ms,
            "counters": dict(zip(sharding.COUNTER_NAMES, [int(v) for v in counters])),
            "parity_checked_scans": picked,
            "backend": args.backend if world > 1 else None,
            "seeds_rank0": [int(sharding.shard_seeds(S, 0)[0]), int(sharding.shard_seeds(S, 0)[-1])],
            "h2d_inclusive_scans_per_s": round(S / (ms_step * 1e-3 + t_h2d), 2),
            "h2d_seconds_per_batch": round(t_h2d, 4),
            "gen_seconds": round(t_gen, 2),
        }
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("kernel") == dom and tr.get("scans_per_launch") == S:
                    out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = "profiles/hbm_traffic.json (rocprofv3 --pmc, corrected per MI355X_MICROARCH.md)"
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg3":
            out["cpu_baseline"] = cpu_baseline(params)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_e2e and args.workload == "cfg3":
            ctx.close()   # the batch context's scratch is not needed any more
            out.update(e2e_callback_path(u, O, params))
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
