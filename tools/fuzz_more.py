"""Extended differential fuzzing on the GPU (not part of the test suite): 2500 further random
unorganised clouds / parameter sets (tests/fuzz.py), HIP path in its production configuration
(float fast paths active) against oracle B: labels, ring, sector and detector stages.
    python tools/fuzz_more.py [first_seed last_seed]   (on the GPU box)
Last runs: seeds 10000..12499 and 20000..29999 (round 1), 30000..33999, 40000..79999 and (after k_ring went z-only) 80000..184999 (round 2, after the ring decision on
cot / position ranking changes), 200000..259999 (round 3, HEAD: one record word per slot, curb lists and interval masks in k_beams, k_ring_table a firing at a
time, empty-tile shortcuts), 300000..329999 (round 3, final: wave-per-sector sort for two-run sectors only -- these unorganised clouds take the
workgroup kernel --, the callback path's short sequence with rerun, messages staged as planes, k_index with wave scans, k_beams in two groups),
400000..414999, 500000..559999, 600000..799999, 900000..919999, 1000000..1199999 (round 4: a third of the clouds with points on the sensor's axis -- NaN azimuths, k_nan_rings --, organised-tile path in k_split),
round 5 (two fifths of the clouds with planar-range ties -- std::sort's order, k_star_ties, both passes --; every tenth case also the published order and the marker
points, where equal azimuths follow the reference's quicksort): see DESIGN.md section 2: 0 mismatches."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, oracles as O, urban_road_filter_amd as u
from fuzz import case
ctx = u.Context(32768, 1)
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 12500):
    (x, y, z), p = case(seed)
    lb, ib, st = O.run_b(x, y, z, p, debug=True)
    ctx.set_params(p)
    lg, ig = ctx.classify_xyz(x, y, z)   # the production path proper
    n = len(x)
    ok = np.array_equal(lg, lb)
    ctx.enable_stage_capture(2)          # the same decisions, ring / sector recorded
    lg2, _ = ctx.classify_xyz(x, y, z)
    ok = ok and np.array_equal(lg2, lb)
    if ib["status"] == 0:
        ok = ok and np.array_equal(ctx.read_stage(u.STAGE_RING, n), st["ring"]) and np.array_equal(ctx.read_stage(u.STAGE_DETECT, n), st["detect"])
        if p.star_shaped_method:
            ok = ok and np.array_equal(ctx.read_stage(u.STAGE_SECTOR, n), st["sector"])
    ctx.enable_stage_capture(0)
    if ib["status"] == 0 and seed % 10 == 0:   # the on-demand outputs (equal azimuths: the reference's quicksort order)
        lg3, _ = ctx.classify_xyz(x, y, z)
        road, curb, prob = ctx.ordered_indices(n)
        ok = ok and np.array_equal(road, st["road_order"]) and np.array_equal(curb, st["curb_order"]) and np.array_equal(prob, st["ring10_order"])
        mg = ctx.marker_points()
        ok = ok and mg.shape == st["marker_pts"].shape and np.array_equal(mg, st["marker_pts"])
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, int((lg != lb).sum()), flush=True)
print("extended fuzz:", bad, "mismatches")
