"""Extended run of tests/fuzz_organised.py on the GPU box (not part of the suite): python tools/fuzz_organised_more.py first last
Last run: see profiles/README.md (round 5)."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import oracles as O
import urban_road_filter_amd as u
from fuzz_organised import case

ctx = u.Context(64 * 2048, 1)
bad = 0
first, last = int(sys.argv[1]), int(sys.argv[2])
for seed in range(first, last):
    (x, y, z), p = case(seed)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx.set_params(p)
    lg, ig = ctx.classify_xyz(x, y, z)
    ok = np.array_equal(lg, lb) and all(getattr(ig, k) == ib[k] for k in ("status", "n_roi", "n_rings", "n_road", "n_curb"))
    if ok and ib["status"] == 0:
        ok = np.array_equal(ctx.read_stage(u.STAGE_DETECT, len(x)), st["detect"])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, int((lg != lb).sum()), flush=True)
print("organised fuzz %d..%d: %d mismatches" % (first, last - 1, bad))
