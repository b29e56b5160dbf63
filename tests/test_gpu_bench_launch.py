"""The N > 1 launch path of bench.py on a 1-GPU box: `--gpus 2 --backend gloo` re-launches under
torch.distributed.run, every rank builds its own context (both on device 0), generates its own
seeds and the run's counters are reduced over the process group -- everything the 8-GPU run does
except RCCL itself (the default backend, nccl, needs one device per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = dict(os.environ, MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]       # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_over_gloo_on_one_device():
    S, K = 48, 2
    d = run_bench("--gpus", "2", "--backend", "gloo", "--scans", str(S), "--steps", str(K), "--warmup", "1",
                  "--no-cpu-baseline", "--parity-scans", "2")
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["scaling"] == "weak"
    c = d["counters"]
    assert c["scans"] == 2 * S * K and c["points_in"] == 2 * S * K * 64 * 2048
    assert c["ok_scans"] == 2 * S * K and 0 < c["curb"] < c["road"] < c["roi_points"] <= c["points_in"]
    assert d["seeds_rank0"] == [1, S]               # rank 1 generates S+1 .. 2S (sharding.shard_seeds)
    assert d["value"] > 0 and d["cpu_baseline"] is None and "e2e_latency_ms" not in d


def test_single_rank_line_has_the_contract_keys():
    d = run_bench("--scans", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--front", "2")   # (mode 1 starts at 192 scans)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "other_configs"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["backend"] is None
    # the other BASELINE configurations, each behind its own parity gate (bench.py: other_configs)
    oc = d["other_configs"]
    assert set(oc) == {"cfg2", "cfg5", "default_roi", "sensor_like", "laser_order", "ring_major", "shuffled", "heterogeneous"}
    for name, e in oc.items():
        assert e["scans_per_s"] > 0 and 0 < e["frac"] < 1 and e["parity_checked_scans"], name
    assert oc["cfg5"]["points_per_scan"] == 128 * 4096 and oc["cfg2"]["scans_per_step"] == 1
    # which storage orders took the fused front end (urf_front.hpp): firing order with the lasers permuted and row-major do, a shuffled cloud does not
    assert oc["laser_order"]["front_scans"] == 32 and oc["ring_major"]["front_scans"] == 32 and oc["shuffled"]["front_scans"] == 0
    assert d["front_scans_per_gpu"] == 32


def test_row_major_workload():
    d = run_bench("--workload", "ring_major", "--scans", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--front", "2")
    assert d["front_scans_per_gpu"] == 32 and d["value"] > 0 and "k_front" in d["kernel_ms"]
    assert "row-major" in d["config"]["workload"]


def test_rccl_with_one_rank():
    """RCCL on the one GPU a test box has: the process group is initialised with backend nccl (= RCCL on ROCm)
    and world size 1, the barrier and both all-reduces run on device tensors, the group is destroyed.  What stays
    unproven without a multi-GPU node: RCCL between ranks."""
    S, K = 32, 2
    d = run_bench("--force-dist", "--scans", str(S), "--steps", str(K), "--warmup", "1", "--no-cpu-baseline", "--no-e2e",
                  "--no-other-configs", "--no-outputs")
    assert d["backend"] == "nccl" and d["n_gpus"] == 1
    c = d["counters"]   # unchanged by the reduction over one rank
    assert c["scans"] == S * K and c["points_in"] == S * K * 64 * 2048 and c["ok_scans"] == S * K
    assert 0 < c["curb"] < c["road"] < c["roi_points"] <= c["points_in"]
