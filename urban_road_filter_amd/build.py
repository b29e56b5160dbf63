"""Builds the in-tree native libraries (gfx950 HIP kernels + C ABI).

    python -m urban_road_filter_amd.build            # liburf_hip.so + liburf_hip_test.so

liburf_hip.so       the product: exactly the entry points of include/urf.h (+ the C++ adapter)
liburf_hip_test.so  the same sources compiled with -DURF_ENABLE_TEST_HOOKS plus synth.cpp: additionally
                    exports include/urf_test_hooks.h (synthetic sweeps, the benchmark's submit / collect
                    loop, device self tests, debug flags).  tests/ and bench.py use it for those calls only.

The shared libraries are written next to this file so that they travel with a
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
LIB_TEST = os.path.join(HERE, "liburf_hip_test.so")

SOURCES = ["urf_api.hip", "params.cpp", "detector.cpp", "marker.cpp"]
HOOK_SOURCES = ["synth.cpp"]                 # liburf_hip_test.so only
HOOK_DEFINES = ["URF_ENABLE_TEST_HOOKS=1"]
HEADERS = ["urf_internal.hpp", "urf_device.hpp", "urf_kernels.hpp", "urf_k_table.hpp", "urf_k_split.hpp", "urf_k_star.hpp", "urf_k_ring.hpp",
           "urf_k_beams_label.hpp", "urf_k_outputs.hpp", "urf_front.hpp", "detector.hpp", "marker.hpp",
           "../../include/urf.h", "../../include/urf_test_hooks.h", "../../include/urf_libm.h"]

# -ffp-contract=off: the reference is built without FMA contraction and label
# parity needs the same roundings; no fast-math anywhere.
# -Bsymbolic: calls between the library's own entry points bind inside the library, so that the product and the
# hooks build can live in one process (tests) without one resolving into the other.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wl,-Bsymbolic",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _compile_lib(out, defines, sources, verbose):
    _run([_hipcc()] + FLAGS + ["-D" + d for d in defines] + sources + ["-o", out], verbose)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    """Both libraries (in parallel); returns the product's path."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hook_srcs = [os.path.join(CSRC, s) for s in HOOK_SOURCES]
    deps = srcs + hook_srcs + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    if force or _stale(LIB, deps):
        jobs.append((LIB, [], srcs))
    if force or _stale(LIB_TEST, deps):
        jobs.append((LIB_TEST, HOOK_DEFINES, srcs + hook_srcs))
    if len(jobs) == 2:   # side by side
        import concurrent.futures as cf
        with cf.ThreadPoolExecutor(2) as ex:
            for f in [ex.submit(_compile_lib, out, d, sr, verbose) for out, d, sr in jobs]:
                f.result()
    else:
        for out, d, sr in jobs:
            _compile_lib(out, d, sr, verbose)
    return LIB


def build_variant(name, defines, verbose=False):
    """A/B builds for kernel experiments: tools/ab/liburf_hip_<name>.so compiled with extra -D flags
    (select with URF_LIB_PATH; bench.py / the tests then run that library for everything: a variant is
    always built with the test hooks)."""
    out_dir = os.path.join(ROOT, "tools", "ab")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "liburf_hip_%s.so" % name)
    srcs = [os.path.join(CSRC, s) for s in SOURCES + HOOK_SOURCES]
    raw = [d for d in defines if d.startswith("-")]   # (compiler flags pass through: -mllvm -align-loops=64 ...)
    _run([_hipcc()] + FLAGS + raw + ["-D" + d for d in HOOK_DEFINES + [d for d in defines if not d.startswith("-")]] + srcs + ["-o", out], verbose)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":   # --variant NAME [DEFINE ...]
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        build(force="--force" in sys.argv)
