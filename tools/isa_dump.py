#!/usr/bin/env python3
"""Device ISA of the library's kernels, one file per kernel under /tmp/isa, with static instruction counts.
usage: python tools/isa_dump.py [kernel ...]   (needs hipcc; compiles urf_api.hip for gfx950 with -S)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/isa"
os.makedirs(OUT, exist_ok=True)
src = os.path.join(ROOT, "urban_road_filter_amd", "csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                       "-fno-gpu-rdc", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + src, "--cuda-device-only", "-S",
                       os.path.join(src, "urf_api.hip"), "-o", os.path.join(OUT, "all.s")] + sys.argv[1:0])
txt = open(os.path.join(OUT, "all.s")).read()
want = sys.argv[1:]
for m in re.finditer(r"^(_Z\d+(k_\w+?)9urf_kargs\w*|_Z\d+(k_\w+?)P\w*):.*?\n(.*?)\n\.Lfunc_end", txt, re.S | re.M):
    name = m.group(2) or m.group(3)
    if want and name not in want:
        continue
    body = m.group(4)
    open(os.path.join(OUT, name + ".s"), "w").write(body)
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    c = lambda p: sum(1 for l in lines if l.startswith(p))
    print("%-22s valu %5d salu %5d ds %4d global %4d scratch %3d" % (name, c("v_"), c("s_"), c("ds_"), c("global_"), c("scratch_")))
