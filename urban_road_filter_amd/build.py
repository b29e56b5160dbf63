"""Builds the in-tree native libraries (gfx950 HIP kernels + C ABI).

    python -m urban_road_filter_amd.build            # liburf_hip.so

The shared library is written next to this file so that it travels with a
snapshot of the repository; nothing is installed into site-packages.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liburf_hip.so")

SOURCES = ["urf_api.hip", "params.cpp", "synth.cpp", "detector.cpp", "marker.cpp"]
HEADERS = ["urf_internal.hpp", "urf_device.hpp", "urf_kernels.hpp", "detector.hpp", "marker.hpp",
           "../../include/urf.h", "../../include/urf_libm.h"]

# -ffp-contract=off: the reference is built without FMA contraction and label
# parity needs the same roundings; no fast-math anywhere.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [_hipcc()] + FLAGS + srcs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_variant(name, defines, verbose=False):
    """A/B builds for kernel experiments: tools/ab/liburf_hip_<name>.so compiled with extra -D flags
    (select with URF_LIB_PATH; bench.py / the tests then run that library)."""
    out_dir = os.path.join(ROOT, "tools", "ab")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liburf_hip_%s.so" % name)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    cmd = [_hipcc()] + FLAGS + ["-D" + d for d in defines] + srcs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":   # --variant NAME [DEFINE ...]
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        build(force="--force" in sys.argv)
