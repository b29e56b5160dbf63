"""Differential fuzzing on CPU: oracle B against the reference's own binary (oracle A) on random
unorganised clouds and random parameters (tests/fuzz.py)."""
import numpy as np
import pytest

import oracles as O
from fuzz import case


@pytest.mark.skipif(not O.has_oracle_a(), reason="oracle A binary (reference build) not available")
@pytest.mark.parametrize("seed", range(100))
def test_oracle_b_equals_reference_on_random_input(seed):
    (x, y, z), p = case(1000 + seed, for_reference=True)
    la, ia, _, _ = O.run_a([(x, y, z)], p)
    lb, ib, _ = O.run_b(x, y, z, p)
    assert ia[0]["status"] == ib["status"]
    assert np.array_equal(la[0], lb & O.MASK_NO_RING), "seed %d: %d labels differ" % (seed, int((la[0] != (lb & O.MASK_NO_RING)).sum()))
    if ib["status"] == 0:   # with < 30 ROI points the reference publishes nothing, not even its roi cloud
        for k in ("n_roi", "n_road", "n_curb", "n_ring10"):
            assert ia[0][k] == ib[k], k
