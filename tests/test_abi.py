"""The C-ABI library loads without a GPU and exports every function include/urf.h declares."""
import ctypes
import os
import sys
import re

import pytest

import urban_road_filter_amd as u

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="urf.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(urf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 18
    L = ctypes.CDLL(u.lib_path())
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_test_hooks_live_in_their_own_library():
    """include/urf_test_hooks.h (synthetic sweeps, the benchmark loop, self tests, debug flags) is exported by
    liburf_hip_test.so only: a node that links the product gets none of it in its symbol table."""
    hooks = declared_functions("urf_test_hooks.h")
    assert set(hooks) == set(u.api.HOOK_SYMBOLS)
    product = ctypes.CDLL(u.lib_path())
    assert not [n for n in hooks if hasattr(product, n)]
    tl = ctypes.CDLL(u.lib_path(hooks=True))
    assert not [n for n in hooks + declared_functions() if not hasattr(tl, n)]


def test_abi_version_and_struct_size():
    assert u.lib().urf_abi_version() == 5   # include/urf.h: URF_ABI_VERSION
    p = u.default_params()
    assert p.size == ctypes.sizeof(u.Params) == 104
    assert ctypes.sizeof(u.ScanInfo) == 32


def test_defaults_are_the_reference_defaults():
    """cfg/LidarFilters.cfg:10-84 + lidar_segmentation.cpp:4 + star_shaped_search.cpp:8-9."""
    p = u.default_params()
    got = {k: getattr(p, k) for k, _ in u.Params._fields_ if k != "size"}
    f = lambda v: ctypes.c_float(v).value  # noqa: E731
    want = dict(x_zero_method=1, z_zero_method=1, star_shaped_method=1, blind_spots=1, xDirection=0,
                interval=f(0.18), curbHeight=f(0.05), curbPoints=5, beamZone=30.0,
                min_X=0.0, max_X=30.0, min_Y=-10.0, max_Y=10.0, min_Z=-3.0, max_Z=-1.0,
                angleFilter1=150.0, angleFilter2=140.0, angleFilter3=50.0,
                kdev_param=f(1.225), kdist_param=2.0, starbeam_filter=0, dmin_param=10,
                channels=64, sectors=360, beam_width=f(0.2))
    assert got == want


def test_strerror_covers_all_codes():
    L = u.lib()
    for code in (0, 1, -1, -2, -3, -4, -5, -6, -7):
        assert L.urf_strerror(code) and b"unknown" not in L.urf_strerror(code)
    assert b"unknown" in L.urf_strerror(-99)


def test_no_cpu_fallback_without_gpu():
    """The product path must fail loudly when there is no device -- it never routes to the oracle."""
    from conftest import gpu_available
    if gpu_available():
        pytest.skip("a GPU is present")
    with pytest.raises(u.UrfError) as e:
        u.Context(1024, 1)
    assert e.value.code == -2


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "urban_road_filter_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.lower(), fn


@pytest.mark.parametrize("client", ["detector_demo", "marker_demo", "two_contexts_demo"])
def test_cpp_clients_build_against_the_product_library(tmp_path, client):
    """What a maintainer's node does with the library -- csrc/detector.hpp and include/urf.h compiled with g++ and linked
    against liburf_hip.so (no hipcc, no test hooks) -- builds here without a GPU; tests/test_gpu_detector.py and
    tests/test_markers.py run the same clients on one."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    u.lib()   # (built)
    pkg = os.path.join(ROOT, "urban_road_filter_amd")
    exe = str(tmp_path / client)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"),
                           os.path.join(ROOT, "tests", "cpp", client + ".cpp"), "-o", exe,
                           "-L" + pkg, "-l:liburf_hip.so", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    assert os.path.exists(exe)
    # the client needs nothing the product library does not export
    undefined = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout.split()
    wanted = {w.split("@")[0] for w in undefined if "urf" in w}          # C entry points and the adapter's C++ members
    exported = set(subprocess.run(["nm", "-D", "--defined-only", u.lib_path()], capture_output=True, text=True).stdout.split())
    assert wanted and wanted <= exported, wanted - exported
    from conftest import gpu_available
    if client == "detector_demo" and not gpu_available():
        # no device: the adapter says so and gives up -- there is no host path behind it
        import struct
        import numpy as np
        cloud = tmp_path / "cloud.bin"
        with open(cloud, "wb") as f:
            f.write(struct.pack("<I", 4096))
            for _ in range(4):
                f.write(np.zeros(4096, np.float32).tobytes())
        r = subprocess.run([exe, str(cloud), str(tmp_path / "out.bin"), "1"], capture_output=True, text=True)
        assert r.returncode != 0 and "no usable HIP device" in (r.stderr + r.stdout)
        assert not os.path.exists(tmp_path / "out.bin")


def test_rccl_shard_client_builds(tmp_path):
    """tests/cpp/shard_demo.cpp (one context + one RCCL communicator per device, ncclAllReduce of the counters through rccl.h)
    compiles and links against the product library and librccl without a GPU; tests/test_gpu_detector.py runs it on one.
    Without a device it reports that and writes nothing."""
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no hipcc / rccl.h")
    u.lib()
    pkg = os.path.join(ROOT, "urban_road_filter_amd")
    exe = str(tmp_path / "shard_demo")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shard_demo.cpp"), "-o", exe, "-L" + pkg, "-l:liburf_hip.so", "-lrccl",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    undefined = subprocess.run(["nm", "-u", exe], capture_output=True, text=True).stdout
    assert "ncclAllReduce" in undefined and "ncclCommInitAll" in undefined and "urf_classify_pc2_async" in undefined
    from conftest import gpu_available
    if not gpu_available():
        import struct
        import numpy as np
        cloud = tmp_path / "cloud.bin"
        with open(cloud, "wb") as f:
            f.write(struct.pack("<I", 4096))
            for _ in range(3):
                f.write(np.zeros(4096, np.float32).tobytes())
        r = subprocess.run([exe, str(tmp_path / "out.bin"), "0", str(cloud)], capture_output=True, text=True)
        assert r.returncode != 0 and "no usable HIP device" in (r.stderr + r.stdout)
        assert not os.path.exists(tmp_path / "out.bin")


def test_graft_entry_build_runs():
    """__graft_entry__.build() -- the driver's "does it build" check -- compiles what is stale and asserts the ABI version it expects."""
    import importlib
    sys.path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    g.build()
