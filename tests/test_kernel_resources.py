"""Register / scratch budget of the gfx950 build (hipcc cross-compiles without a GPU):
the hot kernels must not spill (`.private_segment_fixed_size` = ScratchSize 0) and must stay inside
the register budget their occupancy was tuned for.  Round 1 shipped k_ingest and
k_star_sort_small with 52 / 80 bytes of scratch per lane."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources  # noqa: E402


@pytest.fixture(scope="module")
def table():
    rows = kernel_resources.resources()
    if not rows:
        pytest.skip("hipcc did not report kernel resources")
    return {r["name"]: r for r in rows}


# kernel -> (max VGPRs, waves/SIMD the kernel was tuned for)
BUDGET = {"k_ring_table": (128, 4), "k_split": (85, 6), "k_index": (128, 4), "k_star_sort_small": (80, 6),
          "k_star_walk": (168, 3), "k_ring": (85, 6), "k_ring_general": (128, 4), "k_beams": (64, 6), "k_label": (64, 8),
          "k_star_sort_mid": (64, 8), "k_star_walk_few": (240, 2),
          "k_front": (96, 5), "k_label_front": (64, 8)}   # k_star_walk_few: five waves on an empty device, occupancy is not its concern; k_beams: one workgroup of 768 threads per scan, two of them (24 waves) per CU


@pytest.mark.parametrize("kernel", sorted(BUDGET))
def test_hot_kernels_do_not_spill(table, kernel):
    r = table[kernel]
    assert int(r["ScratchSize [bytes/lane]"]) == 0, r
    vg, occ = BUDGET[kernel]
    assert int(r["VGPRs"]) <= vg and int(r["Occupancy [waves/SIMD]"]) >= occ, r


def test_known_exceptions_are_the_documented_ones(table):
    """k_split_repair / k_split_list are the normally empty loops around k_split's tile body (the scans a speculation got wrong,
    the scans the fused front end handed back); k_star_sort_mid, which used to trade 16 bytes of scratch for 8 waves/SIMD, does
    without since r4."""
    spilling = {k for k, r in table.items() if int(r["ScratchSize [bytes/lane]"]) > 0}
    assert spilling <= {"k_split_repair", "k_split_list", "k_ring_list"}, spilling
