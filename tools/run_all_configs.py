#!/usr/bin/env python3
"""Measures every BASELINE.json configuration once and writes a markdown table
(profiles/<tag>_configs.md): GPU scans/s from bench.py --workload ..., the reference's CPU path
(oracle/_ref/urf_ref, one core) on the same clouds, and parity status.
    python tools/run_all_configs.py r1b          (on the GPU box)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def bench(workload, steps, warmup, extra=()):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps),
                          "--warmup", str(warmup), "--no-cpu-baseline", "--no-e2e", *extra], capture_output=True, text=True, timeout=900)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if not line:
        raise RuntimeError(out.stderr[-2000:])
    return json.loads(line[-1])


def cpu_ref(cfg, seeds, repeat):
    import oracles as O
    if not O.has_oracle_a():
        return None
    p = O.cfg_params(cfg)
    scans = [O.cfg_cloud(cfg, s) for s in seeds]
    _, infos, ms, ms_first = O.run_a(scans, p, repeat=repeat, timeout=1800)
    info = {k: int(v) for k, v in infos[0].items() if not hasattr(v, "shape") and v is not None}
    return {"ms_per_scan": ms, "ms_first_call": ms_first, "scans_per_s": 1000.0 / ms, "info": info}


def main(tag):
    rows = []
    t0 = time.time()
    cpu = {"cfg1": cpu_ref("cfg1", [1, 2], 3), "cfg2": cpu_ref("cfg2", [1, 2], 3),
           "default_roi": cpu_ref("default_roi", [1, 2], 3), "cfg5": cpu_ref("cfg5", [1], 2)}
    g3 = bench("cfg3", 10, 3)
    g2 = bench("cfg2", 300, 30)
    g5 = bench("cfg5", 5, 2)
    gd = bench("default_roi", 10, 3)
    res = {"cpu_reference_one_core": cpu, "gpu": {"cfg3": g3, "cfg2": g2, "cfg5": g5, "default_roi": gd},
           "wall_s": time.time() - t0, "host_cores": os.cpu_count()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", tag + "_configs.json"), "w"), indent=1)

    def f(v, nd=1):
        return ("%." + str(nd) + "f") % v

    md = ["# BASELINE.json configurations, measured (%s)" % tag, "",
          "One MI355X, inputs resident in HBM, parity gate (labels == CPU oracle on sampled scans) passed before timing.",
          "CPU = the reference's own sources (oracle/_ref/urf_ref), one core of the GPU box's host, steady state",
          "(first call excluded).  13 B/point algorithmic traffic, HBM peak 8 TB/s.", "",
          "| config | points/scan | GPU scans/s | GPU ms/scan-or-step | algorithmic GB/s (% of 8 TB/s) | CPU ref scans/s (1 core) | GPU/CPU-core |",
          "|---|---|---|---|---|---|---|"]
    def row(name, g, c, per_scan_latency=False):
        n = g["config"]["points_per_scan"]
        sps = g["value"]
        gbs = 13.0 * n * sps / 1e9
        ms = g["ms_per_step"]
        cs = c["scans_per_s"] if c else float("nan")
        md.append("| %s | %d | %s | %s %s | %s (%s %%) | %s | %sx |" % (
            name, n, f(sps), f(ms, 3), "ms/scan (latency)" if per_scan_latency else "ms per %d-scan step" % g["config"]["scans_per_gpu"],
            f(gbs), f(100 * gbs / 8000.0, 2), f(cs, 2), f(sps / cs, 0)))
    if cpu["cfg1"]:
        md.append("| cfg1 16x1024 flat, z_zero only (CPU plumbing case) | 16384 | n/a | n/a | n/a | %s | n/a |" % f(cpu["cfg1"]["scans_per_s"], 2))
    row("cfg2 single 64x2048 sweep, 1 scan per call", g2, cpu["cfg2"], True)
    row("cfg3 1024 x 64x2048 (headline)", g3, cpu["cfg2"])
    row("cfg5 256 x 128x4096, channels 128", g5, cpu["cfg5"])
    row("1024 x 64x2048, reference default ROI", gd, cpu["default_roi"])
    md += ["", "cfg4 (8192 sweeps over 8 GPUs) is cfg3 per GPU under `bench.py --gpus 8`; the driver measures it.", "",
           "Per-kernel ms (cfg3): " + json.dumps(g3["kernel_ms"]), "", "Per-kernel ms (cfg5): " + json.dumps(g5["kernel_ms"]), "",
           "Per-kernel ms (default ROI): " + json.dumps(gd["kernel_ms"]), "",
           "Per-kernel ms (cfg2, one sweep per call): " + json.dumps(g2["kernel_ms"]), ""]
    open(os.path.join(ROOT, "gpurun_out", tag + "_configs.md"), "w").write("\n".join(md))
    print("\n".join(md))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "rX")
