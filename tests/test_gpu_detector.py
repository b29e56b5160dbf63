"""The C++ adapter (urban_road_filter_amd/csrc/detector.hpp), used the way the reference's ROS
callback would use it: compiled with g++ against the in-tree liburf_hip.so, run on the GPU, and its
four output clouds compared with oracle B."""
import os
import subprocess

import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo(tmp_path):
    exe = str(tmp_path / "detector_demo")
    pkg = os.path.join(ROOT, "urban_road_filter_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "detector_demo.cpp"), "-o", exe,
                           "-L" + pkg, "-l:liburf_hip.so", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_adapter_links_on_cpu(tmp_path):
    """No GPU needed: the adapter's symbols are exported by the shared library."""
    build_demo(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,seed", [(1, 5), (2, 6)])
def test_adapter_clouds_equal_oracle(tmp_path, scene, seed):
    exe = build_demo(tmp_path)
    out = str(tmp_path / "labels.bin")
    r = subprocess.run([exe, "64", "2048", str(scene), str(seed), out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    x, y, z = u.synth_cloud(64, 2048, scene, seed)
    p = O.cfg_params("cfg2")
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    blob = open(out, "rb").read()
    got = np.frombuffer(blob, np.uint8, len(x))
    assert np.array_equal(got, lb & O.MASK_NO_RING)
    road_seq = np.frombuffer(blob, np.uint32, ib["n_road"], len(x))
    assert np.array_equal(road_seq, st["road_order"])   # setReferenceOrder(true): the reference's own order
    assert "road %d curb %d roi %d road_probably %d" % (ib["n_road"], ib["n_curb"], ib["n_roi"], ib["n_ring10"]) in r.stdout
    assert "frame left_os1/os1_lidar" in r.stdout
    assert "pc2 published 1 same_labels 1" in r.stdout   # sensor_msgs/PointCloud2 with a permuted field table
