#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of the gfx950 build (hipcc
-Rpass-analysis=kernel-resource-usage on urf_api.hip; cross-compiles without a GPU).
    python tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-gpu-rdc",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "urban_road_filter_amd", "csrc"),
           "-c", os.path.join(ROOT, "urban_road_filter_amd", "csrc", "urf_api.hip"), "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + list(extra)
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            m2 = re.match(r"_Z(\d+)", name)   # _Z<len><name>...
            if m2:
                name = name[len(m2.group(0)):len(m2.group(0)) + int(m2.group(1))]
            cur = {"name": name}
            out.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return out


if __name__ == "__main__":
    rows = resources(sys.argv[1:])
    print("%-22s %6s %6s %6s %8s %5s %8s" % ("kernel", "VGPRs", "AGPRs", "SGPRs", "scratch", "occ", "LDS"))
    for r in rows:
        print("%-22s %6s %6s %6s %8s %5s %8s" % (r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
                                                 r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
                                                 r.get("LDS Size [bytes/block]")))
