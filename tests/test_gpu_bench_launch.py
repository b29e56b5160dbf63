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
    d = run_bench("--scans", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
