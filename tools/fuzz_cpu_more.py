"""Extended differential fuzzing on the CPU (not part of the test suite): oracle B (oracle/urf_oracle.c) against the reference's
own sources built with the shared libm (oracle/_ref/urf_ref_libm) on further random clouds / parameter sets of tests/fuzz.py --
labels, summaries and the three published orders.  Needs /root/reference at build time only (oracle/Makefile).
    python tools/fuzz_cpu_more.py [first_seed last_seed]
Last runs: seeds 2000..105999 (round 4); 5000000..5031999 and 5100000..5299999 (round 5: two fifths of the clouds with planar-range ties, oracle B
sorting with the restated std::sort): 0 mismatches (profiles/README.md)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracles as O  # noqa: E402
from fuzz import case  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
last = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
bad = 0
for seed in range(first, last):
    (x, y, z), p = case(seed, for_reference=True)
    la, ia, _, _ = O.run_a([(x, y, z)], p, libm=True)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ok = ia[0]["status"] == ib["status"] and np.array_equal(la[0], lb & O.MASK_NO_RING)
    if ok and ib["status"] == 0:
        ok = all(ia[0][k] == ib[k] for k in ("n_roi", "n_road", "n_curb", "n_ring10"))
        ok = ok and all(np.array_equal(ia[0][k], st[k]) for k in ("road_order", "curb_order", "ring10_order"))
    if not ok:
        bad += 1
        print("seed %d differs" % seed, flush=True)
    if (seed - first) % 500 == 499:
        print("... %d cases, %d differ" % (seed - first + 1, bad), flush=True)
print("CPU fuzz %d..%d: %d mismatches" % (first, last - 1, bad))
