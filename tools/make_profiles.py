#!/usr/bin/env python3
"""Turns one tools/profile_round.sh session (gpurun_out/<dir>) into the committed evidence:
    profiles/<tag>_bench.json          the un-profiled bench.py JSON line
    profiles/<tag>_kernel_stats.txt    rocprofv3 --kernel-trace --stats summary
    profiles/<tag>_pmc.txt             per-kernel PMC averages
    profiles/hbm_traffic.json          HBM bytes per launch of the dominant kernel (read by bench.py)
  python tools/make_profiles.py gpurun_out/r1b r1b
HBM traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE counts 64 B per
128 B read request (MI355X_MICROARCH.md, HBM section); confirmed here on k_split, which must read
12 B/point = 1.61e9 B per launch and reports FETCH_SIZE x 1024 = 0.82e9.  WRITE_SIZE needs no
correction (k_split writes 29.5 B/point = 3.96e9 B and reports it)."""
import csv
import glob
import io
import json
import os
import subprocess
import sys
from collections import defaultdict
from contextlib import redirect_stdout

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import pmc_summary  # noqa: E402
import rocprof_summary  # noqa: E402


def main(src, tag):
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    line = [ln for ln in open(os.path.join(src, "bench.json")).read().splitlines() if ln.startswith("{")][-1]
    bench = json.loads(line)
    open(os.path.join(prof, tag + "_bench.json"), "w").write(line + "\n")
    db = glob.glob(os.path.join(src, "trace", "*.db"))[0]
    buf = io.StringIO()
    with redirect_stdout(buf):
        rocprof_summary.main(db)
    open(os.path.join(prof, tag + "_kernel_stats.txt"), "w").write(buf.getvalue().replace(ROOT + "/", ""))
    buf = io.StringIO()
    with redirect_stdout(buf):
        pmc_summary.main(os.path.join(src, "pmc"))
    open(os.path.join(prof, tag + "_pmc.txt"), "w").write(buf.getvalue())
    # HBM traffic of the dominant kernel
    dom = bench["roofline"]["kernel"]
    vals = defaultdict(list)
    for f in glob.glob(os.path.join(src, "pmc", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
                vals[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
    def avg(k, c):
        v = vals.get((k, c), [0.0])
        return sum(v) / len(v)
    pipeline = ("k_ring_table", "k_split", "k_table_repair", "k_split_repair", "k_index", "k_star_sort_small", "k_star_sort_mid", "k_nan_rings",
                "k_star_sort_big", "k_star_ties", "k_star_walk", "k_star_walk_few", "k_ring", "k_beams", "k_label",
                "k_front", "k_front_finish", "k_label_front", "k_split_list", "k_ring_list", "k_label_list", "k_star_sort_runs")   # (r6: the fused front end)
    kernels = sorted({k for k, _ in vals if k in pipeline})
    per = {k: int(2 * avg(k, "FETCH_SIZE") * 1024 + avg(k, "WRITE_SIZE") * 1024) for k in kernels}
    # bench.py's timing slot "k_star_sort" spans k_star_sort_small/mid/big (the latter two run over
    # normally empty work lists); every other slot is one kernel
    doms = [k for k in kernels if (k.startswith("k_star_sort") if dom == "k_star_sort" else k == dom)]
    out = {"kernel": dom, "scans_per_launch": bench["config"]["scans_per_gpu"],
           "hbm_bytes_per_launch": sum(per[k] for k in doms),
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "all_kernels_bytes_per_launch": per, "pipeline_bytes_per_step": sum(per.values()), "tag": tag,
           # wave-level VALU instructions per launch (SQ_INSTS_VALU): bench.py prices them with the issue rates measured by
           # tools/bench_micro/valubench.hip (roofline.valu_issue)
           "valu_insts_per_launch": {k: int(avg(k, "SQ_INSTS_VALU")) for k in kernels},
           "valu_insts_per_step": int(sum(avg(k, "SQ_INSTS_VALU") for k in kernels))}
    json.dump(out, open(os.path.join(prof, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    print("value", bench["value"], "ms/step", bench["ms_per_step"], "dominant", dom, bench["roofline"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
