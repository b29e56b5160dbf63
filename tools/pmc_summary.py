#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc passes collected by tools/pmc_profile.sh.
    python tools/pmc_summary.py gpurun_out/pmc > profiles/<tag>_pmc.txt
FETCH_SIZE is doubled for wide coalesced streams as MI355X_MICROARCH.md prescribes is NOT applied
blindly: both the raw counter (KiB) and the 2x-corrected value are printed."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    data = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            data[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in data for c in data[k]})
    print("# per-dispatch averages (rocprofv3 --pmc), bench cfg3 = 1024 sweeps of 64x2048 per dispatch")
    for k in sorted(data, key=lambda k: -sum(data[k].get("SQ_BUSY_CYCLES", [0]))):
        if k.startswith("__amd") or k.startswith("void at::"):
            continue
        print("\n## " + k)
        for c in counters:
            v = data[k].get(c)
            if v:
                print("  %-24s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
        d = data[k]
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d:
            w = sum(d["SQ_WAVES"]) / len(d["SQ_WAVES"])
            print("  -> per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f VMEM_WR %.0f" % tuple(
                (sum(d[c]) / len(d[c])) / w if c in d else 0 for c in
                ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            fs = sum(d.get("FETCH_SIZE", [0])) / max(1, len(d.get("FETCH_SIZE", [0])))
            ws = sum(d.get("WRITE_SIZE", [0])) / max(1, len(d.get("WRITE_SIZE", [0])))
            print("  -> HBM bytes/dispatch: fetch %.3e (x2 corrected %.3e)  write %.3e" % (fs * 1024, 2 * fs * 1024, ws * 1024))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
