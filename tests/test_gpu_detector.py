"""The C++ adapter (urban_road_filter_amd/csrc/detector.hpp), used the way the reference's ROS
callback would use it: compiled with g++ against the in-tree PRODUCT library liburf_hip.so, run on the GPU, and its
four output clouds compared with oracle B point for point -- x, y, z AND intensity (the reference copies whole
pcl::PointXYZI records into its clouds, lidar_segmentation.cpp:238-242, 354-367)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracles as O
import urban_road_filter_amd as u

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo(tmp_path, name="detector_demo"):
    exe = str(tmp_path / name)
    pkg = os.path.join(ROOT, "urban_road_filter_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "csrc"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L" + pkg, "-l:liburf_hip.so", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def write_cloud(path, x, y, z, intensity):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(x)))
        for a in (x, y, z, intensity):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())


def read_runs(path):
    """-> list of (published, [roi, road, curb, road_probably] as float32 [k, 4] arrays)"""
    blob = open(path, "rb").read()
    pos, runs = 0, []
    while pos < len(blob):
        (pub,) = struct.unpack_from("<I", blob, pos)
        pos += 4
        clouds = []
        for _ in range(4):
            (k,) = struct.unpack_from("<I", blob, pos)
            pos += 4
            clouds.append(np.frombuffer(blob, np.float32, 4 * k, pos).reshape(-1, 4))
            pos += 16 * k
        runs.append((pub, clouds))
    return runs


def test_adapter_links_on_cpu(tmp_path):
    """No GPU needed: the adapter's symbols are exported by the product library."""
    build_demo(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,seed", [(1, 5), (2, 6)])
def test_adapter_clouds_equal_oracle(tmp_path, scene, seed):
    exe = build_demo(tmp_path)
    x, y, z = u.synth_cloud(64, 2048, scene, seed)
    inten = (np.arange(len(x)) % 251).astype(np.float32) * 0.5 + 1.0   # a sensor-like channel, NOT the index
    cloud = str(tmp_path / "cloud.bin")
    out = str(tmp_path / "clouds.bin")
    write_cloud(cloud, x, y, z, inten)
    r = subprocess.run([exe, cloud, out], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    p = O.cfg_params("cfg2")
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    pts = np.stack([x, y, z, inten], axis=1)
    roi = np.nonzero(lb & 4)[0]
    road_in = np.nonzero((lb & 3) == 1)[0]
    curb_in = np.nonzero((lb & 3) == 2)[0]
    prob_in = np.nonzero(lb & 16)[0]
    runs = read_runs(out)
    assert len(runs) == 4 + 6
    # run 0: pcl::PointCloud overload in the reference's own published order
    pub, (c_roi, c_road, c_curb, c_prob) = runs[0]
    assert pub == 1
    assert np.array_equal(c_roi, pts[roi])
    assert np.array_equal(c_road, pts[st["road_order"]])
    assert np.array_equal(c_curb, pts[st["curb_order"]])
    assert np.array_equal(c_prob, pts[st["ring10_order"]])
    # run 1: sensor_msgs/PointCloud2 with a permuted field table (intensity z t x ring y), input order
    for k in [1] + list(range(3, 10)):   # ... the Velodyne driver's layout, and the six sweeps that went through submit() / collect()
        pub, (c_roi, c_road, c_curb, c_prob) = runs[k]
        assert pub == 1
        assert np.array_equal(c_roi, pts[roi]) and np.array_equal(c_road, pts[road_in]), k
        assert np.array_equal(c_curb, pts[curb_in]) and np.array_equal(c_prob, pts[prob_in]), k
    # run 2: 23-byte records without an intensity field: pcl::PointXYZI's default (0)
    bare = pts.copy()
    bare[:, 3] = 0.0
    pub, (c_roi, c_road, c_curb, c_prob) = runs[2]
    assert pub == 1 and np.array_equal(c_roi, bare[roi]) and np.array_equal(c_road, bare[road_in])
    assert np.array_equal(c_curb, bare[curb_in]) and np.array_equal(c_prob, bare[prob_in])
    assert "road %d curb %d roi %d road_probably %d" % (ib["n_road"], ib["n_curb"], ib["n_roi"], ib["n_ring10"]) in r.stdout
    assert "frame left_os1/os1_lidar" in r.stdout
    assert "pc2 published 1 same_labels 1" in r.stdout
    assert "bare published 1 same_labels 1" in r.stdout
    assert "velodyne published 1 same_labels 1" in r.stdout
    assert "pipelined sweeps 6 same_labels 6" in r.stdout


@pytest.mark.gpu
def test_adapter_with_a_narrow_region_of_interest(tmp_path):
    """The reference's default region drops most of a sweep: the label scan skips the empty stretches, the roi cloud is
    no block copy of the message; and the adapter's timing lines parse."""
    exe = build_demo(tmp_path)
    x, y, z = u.synth_cloud(64, 2048, 1, 9)
    inten = np.cos(np.arange(len(x))).astype(np.float32)
    cloud = str(tmp_path / "cloud.bin")
    out = str(tmp_path / "clouds.bin")
    write_cloud(cloud, x, y, z, inten)
    r = subprocess.run([exe, cloud, out, "5", "default_roi"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    p = O.cfg_params("default_roi")
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    pts = np.stack([x, y, z, inten], axis=1)
    runs = read_runs(out)
    pub, (c_roi, c_road, c_curb, c_prob) = runs[0]
    assert pub == 1 and 0 < len(c_roi) < len(x) // 2
    assert np.array_equal(c_roi, pts[np.nonzero(lb & 4)[0]])
    assert np.array_equal(c_road, pts[st["road_order"]]) and np.array_equal(c_curb, pts[st["curb_order"]])
    pub, (c_roi, c_road, c_curb, c_prob) = runs[1]
    assert np.array_equal(c_road, pts[np.nonzero((lb & 3) == 1)[0]]) and np.array_equal(c_prob, pts[np.nonzero(lb & 16)[0]])
    times = dict(line.split()[1:3] for line in r.stdout.splitlines() if line.startswith("time "))
    assert set(times) >= {"pointcloud_input_order", "pointcloud2_permuted_fields", "pointcloud_reference_order",
                          "pointcloud_input_order_with_marker", "pipelined_per_sweep"}
    assert all(float(v) > 0 for v in times.values())


@pytest.mark.gpu
def test_one_context_per_thread_from_cpp(tmp_path):
    """tests/cpp/two_contexts_demo.cpp: two contexts in one process, one host thread each (context i on device i modulo the
    number of devices), four sweeps in flight per context, different parameters per context -- the C-level shape of the
    multi-GPU benchmark (one context per GPU, counters the only thing exchanged).  Labels equal oracle B sweep by sweep."""
    exe = build_demo(tmp_path, "two_contexts_demo")
    clouds = [u.synth_cloud(64, 2048, 1 + k % 2, 200 + k) for k in range(10)]
    files = []
    for k, (x, y, z) in enumerate(clouds):
        files.append(str(tmp_path / ("c%d.bin" % k)))
        with open(files[-1], "wb") as f:
            f.write(struct.pack("<I", len(x)) + x.tobytes() + y.tobytes() + z.tobytes())
    out = str(tmp_path / "labels.bin")
    r = subprocess.run([exe, out] + files, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    blob = np.fromfile(out, np.uint8).reshape(len(clouds), -1)
    road = curb = 0
    for k, (x, y, z) in enumerate(clouds):
        p = O.cfg_params("cfg2")
        if k % 2 == 1:
            p.curbHeight = 0.06
        lb, ib, _ = O.run_b(x, y, z, p)
        assert np.array_equal(blob[k], lb), k
        road += ib["n_road"]
        curb += ib["n_curb"]
    assert "contexts 2 sweeps 5 + 5 road %d curb %d" % (road, curb) in r.stdout


@pytest.mark.gpu
def test_shards_and_rccl_counters_from_cpp(tmp_path):
    """tests/cpp/shard_demo.cpp: the multi-GPU pattern with its collective, from C++ -- one context, one host thread and one
    RCCL communicator per device (ncclCommInitAll), scan s on device s mod G, ncclAllReduce(sum) of the six counters and
    ncclAllReduce(max) of the elapsed time through rccl.h.  The test box has one GPU: G = 1, the collectives run over a
    communicator of one rank.  Labels equal oracle B, the reduced counters equal the sums over the sweeps."""
    exe = str(tmp_path / "shard_demo")
    pkg = os.path.join(ROOT, "urban_road_filter_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shard_demo.cpp"), "-o", exe, "-L" + pkg, "-l:liburf_hip.so", "-lrccl",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"])
    clouds = [u.synth_cloud(64, 2048, 1 + k % 2, 300 + k) for k in range(6)]
    files = []
    for k, (x, y, z) in enumerate(clouds):
        files.append(str(tmp_path / ("c%d.bin" % k)))
        with open(files[-1], "wb") as f:
            f.write(struct.pack("<I", len(x)) + x.tobytes() + y.tobytes() + z.tobytes())
    out = str(tmp_path / "labels.bin")
    r = subprocess.run([exe, out, "0"] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    blob = np.fromfile(out, np.uint8).reshape(len(clouds), -1)
    p = O.cfg_params("cfg2")
    tot = {"n_roi": 0, "n_road": 0, "n_curb": 0}
    for k, (x, y, z) in enumerate(clouds):
        lb, ib, _ = O.run_b(x, y, z, p)
        assert np.array_equal(blob[k], lb), k
        for key in tot:
            tot[key] += ib[key]
    want = "scans %d points_in %d roi_points %d road %d curb %d ok_scans %d" % (len(clouds), len(clouds) * 64 * 2048, tot["n_roi"], tot["n_road"],
                                                                                 tot["n_curb"], len(clouds))
    assert want in r.stdout, r.stdout
