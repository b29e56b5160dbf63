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
The default backend is nccl (= RCCL on ROCm).  --force-dist initialises the process group with ONE rank
and runs the same barrier and all-reduces on device tensors: the RCCL plumbing (communicator, device
tensors, destroy) on the one GPU a test box has.

Rank 0 at N = 1 also measures, each behind its own parity gate and in a few seconds, the other
BASELINE.json configurations (`other_configs`: cfg2 single sweep, cfg5 128x4096, the reference's
default ROI); `value` / `config` / `roofline` are cfg3's.
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
    "sensor": dict(rings=64, cols=2048, scans=1024, params="cfg2", scene=3,
                   text="batch of %d 64x2048 street sweeps as a sensor's driver delivers them (range noise sigma 1 cm, 2 mm range steps, 1.5 %% drop-outs, "
                        "~10 000 planar-range ties per sweep left in), all three detectors + blind_spots, ROI +-200 m"),
    "ring_major": dict(rings=64, cols=2048, scans=1024, params="cfg2", layout="rows",
                       text="batch of %d cfg3 sweeps stored ring by ring (row-major 64 x 2048: an organised cloud), all three detectors + blind_spots, ROI +-200 m"),
    "default_roi": dict(rings=64, cols=2048, scans=1024, params="default_roi",
                        text="batch of %d 64x2048 street sweeps with the reference's DEFAULT ROI (x 0..30, y -10..10: ~40 %% of the points survive)"),
}


def gen_batch(n_scans, seed0, world=1, scene=1):
    """n_scans distinct street sweeps (seeds seed0..), generated on a thread pool
    (urf_synth_cloud releases the GIL); the ranks of a node share its cores."""
    import urban_road_filter_amd as u
    X = np.empty((n_scans, N_PTS), np.float32)
    Y = np.empty_like(X)
    Z = np.empty_like(X)

    def one(s):
        x, y, z = u.synth_cloud(RINGS, COLS, scene, seed0 + s)
        X[s], Y[s], Z[s] = x, y, z

    with cf.ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 1) // max(world, 1)))) as ex:
        list(ex.map(one, range(n_scans)))
    return X, Y, Z


CPU_BASELINE_MAX_PROCS = 64
# the library brackets eight stages of the pipeline with events (urf_kernel_name); when the fused front end takes the batch three of
# the brackets hold its kernels (next to list-driven legacy kernels that find an empty list, ~5 us each)
FRONT_NAMES = {"k_split": "k_front", "k_ring": "k_front_finish", "k_label": "k_label_front"}
FRONT_NOTE = ("fused front end (urf_front.hpp): k_front = k_front + k_table_repair + k_split_list [+ the repair pair]; k_front_finish = k_ring_list + "
              "k_front_finish + k_nan_rings; k_label_front = k_label_list + k_label_front; k_star_sort = k_star_sort_small + the list kernels + "
              "k_star_ties; k_star_walk = k_star_walk + k_star_ties (second pass)")


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
        # P concurrent processes, P swept over {16, 32, 64} (capped by the host's cores): the best total is
        # the all-core figure -- more processes are not more throughput here (page-fault bound, see above)
        sweep = {}
        with tempfile.TemporaryDirectory() as td:
            import struct
            fin = os.path.join(td, "in.bin")
            with open(fin, "wb") as f:
                f.write(b"URFREFIN")
                f.write(struct.pack("<4I", len(scans), N_PTS, 1 + budget_scans // 2, 0))
                f.write(bytes(params))
                for x, y, z in scans:
                    f.write(x.tobytes()); f.write(y.tobytes()); f.write(z.tobytes())
            for P in sorted({min(cores_avail, q) for q in (16, 32, CPU_BASELINE_MAX_PROCS)}):
                procs = [subprocess.Popen([O.ORACLE_A, fin, os.path.join(td, "o%d.bin" % i)]) for i in range(P)]
                for pr in procs:
                    pr.wait()
                rates = []
                for i in range(P):
                    blob = open(os.path.join(td, "o%d.bin" % i), "rb").read()
                    rates.append(1000.0 / struct.unpack_from("<d", blob, 16)[0])
                sweep[P] = float(sum(rates))
        P = max(sweep, key=lambda q: sweep[q])
        sample = ("%d x 64x2048 street sweeps per process after 1 excluded warm-up call; value = sum over P concurrent "
                  "single-threaded processes, best of P in %s: P = %d.  The reference's timed call is its whole Detector::filtered, "
                  "which also holds the marker search and line strips (lidar_segmentation.cpp:295-351, 369-602) that the timed GPU "
                  "step does not: a stated baseline, not a like-for-like ratio" % (2 * (1 + budget_scans // 2) - 1, sorted(sweep), P))
        return {"value": round(sweep[P], 3), "unit": "scans/s", "cores": P, "kind": kind, "sample": sample,
                "single_core_value": round(single, 3), "host_cores_available": cores_avail,
                "value_by_processes": {str(q): round(v, 3) for q, v in sorted(sweep.items())},
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
    with four sweeps in flight (urf_classify_pc2_async: copies and kernels of the sweeps in flight overlap on
    four scratch rows / streams).  A staged message crosses PCIe as x / y / z planes (12 bytes per point); a
    producer that fills the pinned buffer itself sends the records."""
    n = RINGS * COLS
    recs = []
    for k in range(n_sweeps):
        x, y, z = u.synth_cloud(RINGS, COLS, 1, 9000 + k)
        buf = np.zeros((n, 32), np.uint8)
        buf[:, 0:4] = x.view(np.uint8).reshape(-1, 4)
        buf[:, 4:8] = y.view(np.uint8).reshape(-1, 4)
        buf[:, 8:12] = z.view(np.uint8).reshape(-1, 4)
        recs.append(buf.reshape(-1))
    # PCIe bounds at the 56 GB/s large pinned copies reach on the test box (tools/h2d_rate.py; a 4 MiB copy: 50 GB/s)
    out = {"bytes_in_per_scan": int(recs[0].nbytes), "bytes_out_per_scan": n, "bytes_over_pcie_staged": 12 * n + n,
           "pcie_bound_scans_per_s": round(56e9 / (recs[0].nbytes + n), 1),
           "pcie_bound_scans_per_s_staged": round(56e9 / (12 * n + n), 1)}
    IN_FLIGHT = 4   # URF_MAX_IN_FLIGHT
    out["e2e_sweeps_in_flight"] = IN_FLIGHT
    # ctx: the product library (liburf_hip.so) -- latency and the Python-client streams; hctx: the hooks build of the same
    # sources (liburf_hip_test.so), which alone exports the library-side submit / collect loop and the debug flags
    with u.Context(n, IN_FLIGHT, params=params) as ctx, u.Context(n, IN_FLIGHT, params=params, hooks=True) as hctx:
        lab = np.empty(n, np.uint8)
        lb, _, _ = O.run_b(*u.synth_cloud(RINGS, COLS, 1, 9000), params)
        for c in (ctx, hctx):
            lg, _ = c.classify_pc2(recs[0], n, 32, 0, 4, 8)
            if not np.array_equal(lg, lb):
                raise SystemExit("parity failure on the callback path")

        def latency(c, flags=0):
            if flags:
                c.set_debug_flags(flags)
            for k in range(4):
                c.classify_pc2(recs[k % n_sweeps], n, 32, 0, 4, 8)
            ts = []
            for k in range(reps):
                t0 = time.perf_counter()
                t = c.classify_pc2_async(recs[k % n_sweeps], n, 32, 0, 4, 8)
                c.classify_pc2_wait(t, lab)
                ts.append(time.perf_counter() - t0)
            if flags:
                c.set_debug_flags(0)
            return 1e3 * float(np.median(ts))

        out["e2e_latency_ms"] = round(latency(ctx), 4)                        # graph replay, product library
        out["e2e_latency_ms_kernel_by_kernel"] = round(latency(hctx, 8), 4)   # the same without the captured graph (debug flag: hooks build)

        def stream(zero_copy):
            inflight = []
            t0 = time.perf_counter()
            for k in range(stream_reps):
                if len(inflight) == IN_FLIGHT:
                    ctx.classify_pc2_wait(inflight.pop(0), lab)
                rec = recs[k % n_sweeps]
                if zero_copy:   # the producer (a driver, a deserialiser) fills the pinned buffer itself
                    pin = ctx.pinned_input(rec.nbytes)
                    if k < IN_FLIGHT:   # (the bench writes each of the slots' buffers once: producing the data is not what is timed)
                        pin[:] = rec
                    inflight.append(ctx.classify_pc2_async(pin.ctypes.data, n, 32, 0, 4, 8))
                else:
                    inflight.append(ctx.classify_pc2_async(rec, n, 32, 0, 4, 8))
            for t in inflight:
                ctx.classify_pc2_wait(t, lab)
            return stream_reps / (time.perf_counter() - t0)

        # (the slots' sequences are captured again when the message format changes, and with a pinned producer the host
        # thread mostly sleeps in the wait: a first run of 160 sweeps measures the wake-up of host and device -- one
        # untimed run, then the better of two)
        stream(False)
        out["e2e_overlapped_scans_per_s_python_client"] = round(max(stream(False), stream(False)), 1)
        stream(True)
        out["e2e_overlapped_scans_per_s_pinned_producer_python_client"] = round(max(stream(True), stream(True)), 1)
        # the same loops inside the library (urf_bench_callback_stream): what a C / C++ client -- the reference is a
        # C++ node -- gets, without a Python interpreter between the calls; the labels of the last sweep are checked
        # (the better of two passes, like the Python client's figures above: the host's side of the staged path -- a gather of
        # 4 MiB per sweep -- varies with whatever else the box's cores are doing)
        hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, 16, IN_FLIGHT)
        lbn, _, _ = O.run_b(*u.synth_cloud(RINGS, COLS, 1, 9000 + (stream_reps - 1) % n_sweeps), params)
        best = None
        for _ in range(2):
            sec, labn = hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, stream_reps, IN_FLIGHT)
            if not np.array_equal(labn, lbn):
                raise SystemExit("parity failure on the callback path (native loop)")
            best = sec if best is None else min(best, sec)
        out["e2e_overlapped_scans_per_s"] = round(stream_reps / best, 1)
        hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, 16, IN_FLIGHT, producer_pinned=True)
        best = min(hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, stream_reps, IN_FLIGHT, producer_pinned=True)[0] for _ in range(2))
        out["e2e_overlapped_scans_per_s_pinned_producer"] = round(stream_reps / best, 1)
        hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, 8, 1)
        sec, _ = hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, reps, 1)
        out["e2e_latency_ms_native_mean"] = round(1e3 * sec / reps, 4)
        # the same sweeps with the reference's DEFAULT region of interest (cfg/LidarFilters.cfg:42-51: what a node that
        # switches libraries runs): fewer points to classify, but the ring table takes its late leaders one at a time
        p_roi = O.cfg_params("default_roi")
        lbr, _, _ = O.run_b(*u.synth_cloud(RINGS, COLS, 1, 9000), p_roi)
        for c in (ctx, hctx):
            c.set_params(p_roi)
            lgr, _ = c.classify_pc2(recs[0], n, 32, 0, 4, 8)
            if not np.array_equal(lgr, lbr):
                raise SystemExit("parity failure on the callback path (default ROI)")
        out["e2e_latency_ms_default_roi"] = round(latency(ctx), 4)
        hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, 16, IN_FLIGHT)
        sec = min(hctx.bench_callback_stream(recs, n, 32, 0, 4, 8, stream_reps, IN_FLIGHT)[0] for _ in range(2))
        out["e2e_overlapped_scans_per_s_default_roi"] = round(stream_reps / sec, 1)
        # a sweep as a sensor's driver delivers it (planar-range ties in every star sector): both passes of k_star_ties are part of
        # the launch sequence after the first such sweep (urf_callback_path_state bit 4)
        xs, ys, zs = u.synth_cloud(RINGS, COLS, 3, 9000)
        bufs = np.zeros((n, 32), np.uint8)
        bufs[:, 0:4] = xs.view(np.uint8).reshape(-1, 4)
        bufs[:, 4:8] = ys.view(np.uint8).reshape(-1, 4)
        bufs[:, 8:12] = zs.view(np.uint8).reshape(-1, 4)
        recs_s = [bufs.reshape(-1)]
        lbs, _, _ = O.run_b(xs, ys, zs, params)
        ctx.set_params(params)
        lgs, _ = ctx.classify_pc2(recs_s[0], n, 32, 0, 4, 8)
        if not np.array_equal(lgs, lbs):
            raise SystemExit("parity failure on the callback path (sensor-like sweep)")
        recs, keep = recs_s * n_sweeps, recs
        out["e2e_latency_ms_sensor_like"] = round(latency(ctx), 4)
        recs = keep
    # a row-major organised sweep (height = the lasers: an Ouster's cloud) on the callback path, in a context of its own: the first sweep
    # sights the layout (general kernels), the following ones take the fused kernels inside the captured sequences (r6)
    with u.Context(n, IN_FLIGHT, params=params) as rctx:
        xr, yr, zr = (np.ascontiguousarray(a.reshape(COLS, RINGS).T.reshape(-1)) for a in u.synth_cloud(RINGS, COLS, 1, 9100))
        bufr = np.zeros((n, 32), np.uint8)
        bufr[:, 0:4] = xr.view(np.uint8).reshape(-1, 4)
        bufr[:, 4:8] = yr.view(np.uint8).reshape(-1, 4)
        bufr[:, 8:12] = zr.view(np.uint8).reshape(-1, 4)
        lbr2, _, _ = O.run_b(xr, yr, zr, params)
        for k in range(4):
            lgr2, _ = rctx.classify_pc2(bufr.reshape(-1), n, 32, 0, 4, 8)
            if not np.array_equal(lgr2, lbr2):
                raise SystemExit("parity failure on the callback path (row-major sweep, call %d)" % k)
        recs, keep = [bufr.reshape(-1)] * n_sweeps, recs
        out["e2e_latency_ms_row_major"] = round(latency(rctx), 4)
        out["e2e_row_major_fused"] = rctx.front_scans()
        recs = keep
    out["e2e_method"] = ("every e2e_overlapped_* figure: one untimed pass, then the better of two timed passes of %d sweeps; e2e_latency_ms*: median of %d "
                         "synchronous calls after 4 untimed ones (e2e_latency_ms_native_mean: mean of %d).  Library: e2e_latency_ms, *_default_roi latency and the "
                         "*_python_client streams run in liburf_hip.so (the product); the native-loop streams, e2e_latency_ms_native_mean and "
                         "e2e_latency_ms_kernel_by_kernel in liburf_hip_test.so (same sources + include/urf_test_hooks.h, which alone exports the loop)"
                         % (stream_reps, reps, reps))
    return out


def adapter_e2e(u, reps=40):
    """The unit the north-star names, through the piece a maintainer links: urf::Detector::filtered
    (urban_road_filter_amd/csrc/detector.hpp) -- one 64x2048 sweep as a pcl::PointCloud / sensor_msgs::PointCloud2 in
    host memory in, the four clouds road / curb / roi / road_probably in host memory out
    (lidar_segmentation.cpp:95 -> :354-367, 605-621).  tests/cpp/detector_demo.cpp, compiled with g++ against the product
    library, times the call (median of `reps`); ROI +-200 m, i.e. all 131 072 points reach the roi cloud."""
    import struct
    pkg = os.path.join(ROOT, "urban_road_filter_amd")
    try:
        with tempfile.TemporaryDirectory() as td:
            exe = os.path.join(td, "detector_demo")
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"),
                                   os.path.join(ROOT, "tests", "cpp", "detector_demo.cpp"), "-o", exe,
                                   "-L" + pkg, "-l:liburf_hip.so", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
            x, y, z = u.synth_cloud(RINGS, COLS, 1, 9000)
            inten = (np.arange(len(x)) % 251).astype(np.float32)
            with open(os.path.join(td, "cloud.bin"), "wb") as f:
                f.write(struct.pack("<I", len(x)) + x.tobytes() + y.tobytes() + z.tobytes() + inten.tobytes())
            r = subprocess.run([exe, os.path.join(td, "cloud.bin"), os.path.join(td, "out.bin"), str(reps)], capture_output=True,
                               text=True, timeout=300)
            if r.returncode != 0 or "same_labels 1" not in r.stdout:
                return {"error": (r.stderr or r.stdout)[-300:]}
            t = {ln.split()[1]: float(ln.split()[2]) for ln in r.stdout.splitlines() if ln.startswith("time ")}
            return {"what": "urf::Detector::filtered, message in host memory -> four clouds in host memory, median ms per sweep "
                            "(g++ client of liburf_hip.so; pipelined: submit() / collect() with four sweeps in flight)", **t}
    except Exception as e:   # (no compiler on the box, ...): the bench line does not depend on it
        return {"error": repr(e)[-300:]}


def _timed_steps(torch, stream, fn, steps, warmup):
    """ms per call of fn (asynchronous on `stream`): wall clock around `steps` calls, device synchronised on both sides."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def other_configs(u, O, torch, ctx, stream, dev, dx, dy, dz, dl, di, X, Y, Z, S):
    """The remaining BASELINE.json configurations that fit one GPU, each behind its own parity gate (labels ==
    CPU oracle on sampled scans) and timed like the headline (inputs resident, whole pipeline per step):
      default_roi  the SAME 1024 sweeps with the reference's default region of interest (cfg/LidarFilters.cfg:42-51)
      cfg2         ONE 64x2048 sweep per call (resident): the reference's unit of work, latency
      cfg5         256 x 128x4096 sweeps, channels 128, interval 0.05 (LDS-pressure stress)
    `frac` = 13 B/point x points/s / 8 TB/s, the whole pipeline."""
    res = {}

    def entry(n_pts, n_scans, ms, kms, kcalls, picked, note, fused=None):
        sps = n_scans / (ms * 1e-3)
        return {"scans_per_s": round(sps, 2), "ms_per_step": round(ms, 4), "scans_per_step": n_scans, "points_per_scan": n_pts,
                "frac": round(13.0 * n_pts * sps / (HBM_PEAK_GBS * 1e9), 5),
                "kernel_ms": {k: round(v / max(kcalls, 1), 4) for k, v in kms.items()},
                "front_scans": fused,   # scans of a step that took the fused front end (urf_front.hpp)
                "parity_checked_scans": picked, "workload": note}

    def run(c, fn, n_pts, n_scans, steps, warmup, note, picked):
        ms = _timed_steps(torch, stream, fn, steps, warmup)
        c.enable_kernel_timing(True)
        c.kernel_timing()
        c.enable_kernel_timing(True)
        for _ in range(3):
            fn()
        kms, kcalls = c.kernel_timing()
        c.enable_kernel_timing(False)
        return entry(n_pts, n_scans, ms, kms, kcalls, picked, note, c.front_scans())

    def gate(labels_of, clouds, params, picked, what):
        for s in picked:
            lb, _, _ = O.run_b(*clouds(s), params)
            if not np.array_equal(labels_of(s), lb):
                raise SystemExit("parity failure (%s) on scan %d" % (what, s))

    # ---- default ROI: same inputs, same context, the reference's own region of interest
    p_roi = O.cfg_params("default_roi")
    ctx.set_params(p_roi)
    fn = lambda: ctx.classify_batch_soa(dx, dy, dz, N_PTS, S, dl, di)   # noqa: E731
    fn()
    torch.cuda.synchronize()
    picked = sorted(np.random.default_rng(7).choice(S, min(2, S), replace=False).tolist())
    gate(lambda s: dl[s].cpu().numpy(), lambda s: (X[s], Y[s], Z[s]), p_roi, picked, "default_roi")
    res["default_roi"] = run(ctx, fn, N_PTS, S, 10, 2, WORKLOADS["default_roi"]["text"] % S, picked)
    # ---- cfg2: one resident sweep per call
    p2 = O.cfg_params("cfg2")
    ctx.set_params(p2)
    with u.Context(N_PTS, 1, device=dev.index, params=p2) as c1:
        c1.set_stream(stream.cuda_stream)
        l1 = torch.empty(N_PTS, dtype=torch.uint8, device=dev)
        i2 = min(5, S - 1)
        fn1 = lambda: c1.classify_batch_soa(dx[i2], dy[i2], dz[i2], N_PTS, 1, l1, None)   # noqa: E731
        fn1()
        torch.cuda.synchronize()
        gate(lambda s: l1.cpu().numpy(), lambda s: (X[s], Y[s], Z[s]), p2, [i2], "cfg2")
        res["cfg2"] = run(c1, fn1, N_PTS, 1, 300, 30, WORKLOADS["cfg2"]["text"] % 1, [i2])
        res["cfg2"]["latency_ms_resident"] = res["cfg2"]["ms_per_step"]
    # ---- cfg5: 256 x 128x4096
    R5, C5, S5 = 128, 4096, (256 if S >= 1024 else max(4, S // 4))
    n5 = R5 * C5
    p5 = O.cfg_params("cfg5")
    X5 = np.empty((S5, n5), np.float32)
    Y5 = np.empty_like(X5)
    Z5 = np.empty_like(X5)

    def one(s):
        X5[s], Y5[s], Z5[s] = u.synth_cloud(R5, C5, 1, 1 + s)

    with cf.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(one, range(S5)))
    ex5, ey5, ez5 = (torch.from_numpy(a).to(dev) for a in (X5, Y5, Z5))
    l5 = torch.empty((S5, n5), dtype=torch.uint8, device=dev)
    with u.Context(n5, S5, device=dev.index, params=p5) as c5:
        c5.set_stream(stream.cuda_stream)
        fn5 = lambda: c5.classify_batch_soa(ex5, ey5, ez5, n5, S5, l5, None)   # noqa: E731
        fn5()
        torch.cuda.synchronize()
        picked5 = sorted(np.random.default_rng(11).choice(S5, 2, replace=False).tolist())
        gate(lambda s: l5[s].cpu().numpy(), lambda s: (X5[s], Y5[s], Z5[s]), p5, picked5, "cfg5")
        res["cfg5"] = run(c5, fn5, n5, S5, 5, 2, WORKLOADS["cfg5"]["text"] % S5, picked5)
    del ex5, ey5, ez5, l5
    # ---- sensor-like sweeps: what a driver delivers (range noise, 2 mm range steps, drop-outs) -- ~10 000 exact planar-range
    # ties per sweep, in EVERY star sector, whose order is libstdc++'s std::sort's (star_shaped_search.cpp:109): every sector
    # is sorted a second time by k_star_ties.  The headline's clouds are tie-free by SURVEY.md 8d's rule.
    Ss = S   # (r6: as many as the headline's step, so that the two rates compare)
    Xs = np.empty((Ss, N_PTS), np.float32)
    Ys = np.empty_like(Xs)
    Zs = np.empty_like(Xs)

    def one_s(s):
        Xs[s], Ys[s], Zs[s] = u.synth_cloud(RINGS, COLS, 3, 1 + s)

    with cf.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
        list(ex.map(one_s, range(Ss)))
    sx, sy, sz = (torch.from_numpy(a).to(dev) for a in (Xs, Ys, Zs))
    ctx.set_params(p2)
    fns = lambda: ctx.classify_batch_soa(sx, sy, sz, N_PTS, Ss, dl, di)   # noqa: E731
    fns()
    torch.cuda.synchronize()
    pickeds = sorted(np.random.default_rng(13).choice(Ss, min(4, Ss), replace=False).tolist())
    gate(lambda s: dl[s].cpu().numpy(), lambda s: (Xs[s], Ys[s], Zs[s]), p2, pickeds, "sensor_like")
    res["sensor_like"] = run(ctx, fns, N_PTS, Ss, 10, 2, "batch of %d 64x2048 street sweeps as a sensor's driver delivers them: range noise "
                             "(sigma 1 cm), 2 mm range steps, 1.5 %% drop-outs, ~10 000 planar-range ties per sweep left in (every star sector "
                             "is sorted again in std::sort's order by k_star_ties), ROI +-200 m" % Ss, pickeds)
    del sx, sy, sz
    # ---- the SAME cfg3 sweeps in the storage orders other drivers deliver (r6; the reference makes no assumption about the order,
    # lidar_segmentation.cpp:100-117, 221-278).  laser_order: firing by firing, the 64 lasers of a firing in a fixed non-monotone
    # order (a Velodyne's laser numbering) -- the fused front end learns the lane -> ring map; ring_major: row-major H x W, one ring
    # after the other (an Ouster's organised cloud) -- sighted by the first call, k_transpose + the fused kernels from the second;
    # shuffled: not a sequence of firings, the general kernels take it.
    # Equal input sets: labels are those of the firing-order sweep permuted, gated against oracle B on the permuted input.
    def layout(name, index_of, note):
        idx = torch.from_numpy(index_of.astype(np.int64)).to(dev)
        lx, ly, lz = (t.index_select(1, idx).contiguous() for t in (dx, dy, dz))
        torch.cuda.synchronize()   # (torch's own stream has written them; the library launches on `stream`)
        ctx.set_params(p2)
        fnl = lambda: ctx.classify_batch_soa(lx, ly, lz, N_PTS, S, dl, di)   # noqa: E731
        fnl()
        torch.cuda.synchronize()
        pk = sorted(np.random.default_rng(17).choice(S, min(2, S), replace=False).tolist())
        gate(lambda s: dl[s].cpu().numpy(), lambda s: (X[s][index_of], Y[s][index_of], Z[s][index_of]), p2, pk, name)
        res[name] = run(ctx, fnl, N_PTS, S, 10, 2, note % S, pk)
        torch.cuda.synchronize()   # ... and the labels of the timed calls (a row-major sweep's first call only sights the layout)
        gate(lambda s: dl[s].cpu().numpy(), lambda s: (X[s][index_of], Y[s][index_of], Z[s][index_of]), p2, pk, name + ", timed calls")

    rng_l = np.random.default_rng(23)
    perm64 = rng_l.permutation(RINGS)
    fir = np.arange(N_PTS).reshape(COLS, RINGS)
    layout("laser_order", fir[:, perm64].reshape(-1), "the %d cfg3 sweeps with the 64 lasers of every firing in a fixed random order (laser-number order)")
    layout("ring_major", fir.T.reshape(-1), "the %d cfg3 sweeps stored ring by ring (row-major 64 x 2048: an organised cloud as an Ouster's driver delivers it)")
    layout("shuffled", rng_l.permutation(N_PTS), "the %d cfg3 sweeps with their points in one random order (an unorganised cloud)")
    # ---- a batch in which every 8th sweep defeats the speculative ring table (r6): the reference's default region of interest, every
    # 8th sweep stored from the rear -- no point of its first 8192 lies in the region, k_ring_table gives up with an empty table,
    # the scan is repaired and split again inside the call (and handed back by the fused front end).  first_call_ms: the call that
    # pays for the failed speculation; ms_per_step: the calls after it (the context has stopped speculating).
    with u.Context(N_PTS, S, device=dev.index, params=p_roi) as ch:
        ch.set_stream(stream.cuda_stream)
        rear = torch.arange(0, S, 8, device=dev)
        roll = torch.from_numpy(np.roll(fir, COLS // 2, axis=0).reshape(-1).astype(np.int64)).to(dev)
        hx, hy, hz = (t.clone() for t in (dx, dy, dz))
        for t in (hx, hy, hz):
            t[rear] = t[rear].index_select(1, roll)
        torch.cuda.synchronize()
        fnh = lambda: ch.classify_batch_soa(hx, hy, hz, N_PTS, S, dl, di)   # noqa: E731
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fnh()
        e1.record(stream)
        torch.cuda.synchronize()
        first_ms = e0.elapsed_time(e1)
        fused_first = ch.front_scans()
        roll_np = np.roll(fir, COLS // 2, axis=0).reshape(-1)
        pk = [0, 8, 5] if S > 8 else [0]
        gate(lambda s: dl[s].cpu().numpy(), lambda s: ((X[s][roll_np], Y[s][roll_np], Z[s][roll_np]) if s % 8 == 0 else (X[s], Y[s], Z[s])),
             p_roi, pk, "heterogeneous")
        res["heterogeneous"] = run(ch, fnh, N_PTS, S, 10, 2, "batch of %d default-ROI sweeps, every 8th stored from the rear: the speculative ring table "
                                   "of those fails (repaired in the call; the fused front end hands them back)" % S, pk)
        res["heterogeneous"]["first_call_ms"] = round(first_ms, 4)
        res["heterogeneous"]["front_scans_first_call"] = fused_first
        del hx, hy, hz
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scans", type=int, default=0, help="sweeps per GPU per step (default: the workload's, cfg3: 1024)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-scans", type=int, default=32)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=os.environ.get("URF_BENCH_BACKEND", "nccl"))
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-outputs", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--front", type=int, default=-1, help="urf_set_front_mode: 0 legacy kernels only, 1 / 2 the fused front end (default: the library's)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (backend as given) even with one rank and run its barrier / all-reduces")
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
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit("LOCAL_RANK %d but only %d device(s): RCCL needs one device per rank (--backend gloo shares devices)"
                         % (local_rank, n_dev))
    dev_index = local_rank if args.backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
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
    X, Y, Z = gen_batch(S, sharding.shard_seeds(S, rank)[0], world, wl.get("scene", 1))   # seeds 1..S on rank 0, S+1..2S on rank 1, ...
    if wl.get("layout") == "rows":   # firing order -> row-major: point l * COLS + f
        X, Y, Z = (np.ascontiguousarray(A.reshape(S, COLS, RINGS).transpose(0, 2, 1)).reshape(S, N_PTS) for A in (X, Y, Z))
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
    if args.front >= 0:
        ctx.set_front_mode(args.front)

    def step():
        ctx.classify_batch_soa(dx, dy, dz, N_PTS, S, dl, di)

    # parity gate: no number is reported for a batch whose labels differ from the CPU oracle
    if wl.get("layout") == "rows":   # (a context's first call only sights the layout: the gate is on the sequence that is timed)
        step()
        torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    front_scans = ctx.front_scans()   # scans of the batch that took the fused front end (urf_front.hpp)
    picked = sorted(np.random.default_rng(20260925 + rank).choice(S, min(args.parity_scans, S), replace=False).tolist())
    for s in picked:   # a seeded random sample of the batch, not its first scans
        lb, _, _ = O.run_b(X[s], Y[s], Z[s], params)
        L = dl[s].cpu().numpy()
        if not np.array_equal(L, lb):
            raise SystemExit("parity failure on scan %d of rank %d: %d labels differ" % (s, rank, int((L != lb).sum())))

    for _ in range(args.warmup):
        step()

    def barrier():
        if use_dist:
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
    # per-kernel times come from a SECOND pass of the same K steps with the library's event brackets on (hipEvent pairs
    # around every kernel on the launch stream): `value` above is from a pass without them
    ctx.enable_kernel_timing(True)
    ctx.kernel_timing()                     # reset
    ctx.enable_kernel_timing(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
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
        if front_scans == S:   # every scan took the fused front end: the three brackets hold its kernels (urf_front.hpp)
            kms = {FRONT_NAMES.get(k, k): v for k, v in kms.items()}
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
                       "scans_per_gpu": S, "points_per_scan": N_PTS,
                       "sharding": "one batch per GPU (rank r classifies the contiguous seed block r*S+1 .. (r+1)*S), no data-path collective"},
            "hbm_roofline_frac_whole_pipeline": round(whole / HBM_PEAK_GBS, 5),   # 13 B/point x points/s over the WHOLE step / 8 TB/s: the honest figure
            "roofline": {"bound": "hbm", "whole_pipeline_frac": round(whole / HBM_PEAK_GBS, 5), "whole_pipeline_achieved": round(whole, 2),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "frac_note": "achieved / frac are the contract's: the WHOLE algorithm's bytes over the DOMINANT kernel's time; "
                                      "whole_pipeline_frac divides the same bytes by the whole step and is the figure to compare with the north-star's 40 %",
                         "kernel": dom, "kernel_ms": round(dom_ms, 4),
                         "kernel_ms_source": "hipEvent pairs around the kernel on the launch stream, mean over the timed steps "
                                             "(rocprofv3 --kernel-trace of the same command: profiles/*_kernel_stats.txt)",
                         "algorithmic_bytes_per_launch": alg_bytes_launch},
            "kernel_ms": {k: round(v / max(kcalls, 1), 4) for k, v in kms.items()},
            "kernel_ms_note": FRONT_NOTE if front_scans else None,
            "outputs_ms_per_batch": outputs_ms,
            "counters": dict(zip(sharding.COUNTER_NAMES, [int(v) for v in counters])),
            "front_scans_per_gpu": front_scans,
            "parity_checked_scans": picked,
            "backend": dist.get_backend() if use_dist else None,
            "seeds_rank0": [int(sharding.shard_seeds(S, 0)[0]), int(sharding.shard_seeds(S, 0)[-1])],
            "h2d_inclusive_scans_per_s": round(S / (ms_step * 1e-3 + t_h2d), 2),
            "h2d_seconds_per_batch": round(t_h2d, 4),
            "gen_seconds": round(t_gen, 2),
        }
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("scans_per_launch") == S and tr.get("pipeline_bytes_per_step"):
                    out["roofline"]["traffic_pipeline"] = tr["pipeline_bytes_per_step"]   # all kernels of one step
                if tr.get("kernel") == dom and tr.get("scans_per_launch") == S:
                    out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = "profiles/hbm_traffic.json (rocprofv3 --pmc, corrected per MI355X_MICROARCH.md)"
                if tr.get("valu_insts_per_step") and tr.get("scans_per_launch") == S:
                    # The other ceiling: vector-instruction issue.  Measured on this chip (tools/bench_micro/valubench.hip,
                    # profiles/r4_valubench.txt): a SIMD issues 1.05 G wave64 instructions/s of the two-operand kind (2.3 cycles
                    # at 2.4 GHz -- MI355X_MICROARCH.md's "two cycles", not the four DESIGN.md assumed until r3) and 0.57-0.63 G/s
                    # of the three-operand / compare / 64-bit kind (4 cycles); 256 CUs x 4 SIMDs.
                    n_inst, simds = float(tr["valu_insts_per_step"]), 1024.0
                    fast, slow = 1e3 * n_inst / (simds * 1.05e9), 1e3 * n_inst / (simds * 0.60e9)
                    out["roofline"]["valu_issue"] = {
                        "insts_per_step": int(n_inst), "insts_source": "profiles/hbm_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU, summed over the pipeline's kernels)",
                        "rate_source": "profiles/r4_valubench.txt: 1.05e9 (two-operand VALU) .. 0.60e9 (three-operand, compare, f64) wave-instructions/s/SIMD",
                        "ceiling_ms": round(fast, 4), "ceiling_ms_all_slow_encodings": round(slow, 4),
                        "frac": round(fast / ms_step, 4), "frac_all_slow_encodings": round(slow / ms_step, 4)}
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg3":
            out["cpu_baseline"] = cpu_baseline(params)
        else:
            out["cpu_baseline"] = None
        if world == 1 and args.workload == "cfg3" and not args.no_other_configs:
            # (the outputs above read ring-sorted intermediate results: the context ran the batch again through the general kernels and
            # would stay with them -- include/urf.h; the configurations below are timed as a caller without those outputs sees them)
            ctx.set_front_mode(args.front if args.front >= 0 else 1)
            out["other_configs"] = other_configs(u, O, torch, ctx, stream, dev, dx, dy, dz, dl, di, X, Y, Z, S)
        if world == 1 and not args.no_e2e and args.workload == "cfg3":
            ctx.close()   # the batch context's scratch is not needed any more
            e2e = e2e_callback_path(u, O, params)
            out.update(e2e)
            ad = adapter_e2e(u)
            out["adapter_e2e"] = ad
            out["adapter_e2e_ms"] = ad.get("pointcloud_input_order")
            if "other_configs" in out:
                out["other_configs"]["cfg2"]["e2e_latency_ms"] = e2e["e2e_latency_ms"]
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
