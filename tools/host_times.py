"""Host-side cost of the callback path, phase by phase (URF_HOST_TIMES=1 makes urf_bench_callback_stream
print it): python tools/host_times.py [sweeps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["URF_HOST_TIMES"] = "1"
import numpy as np  # noqa: E402

import urban_road_filter_amd as u  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    n = 64 * 2048
    recs = []
    for seed in range(1, 9):
        x, y, z = u.synth_cloud(64, 2048, 1, seed)
        r = np.zeros((n, 8), np.float32)
        r[:, 0], r[:, 1], r[:, 2] = x, y, z
        recs.append(r.view(np.uint8).reshape(-1).copy())
    p = u.default_params()
    p.min_X, p.max_X, p.min_Y, p.max_Y = -200.0, 200.0, -200.0, 200.0
    with u.Context(n, 4, params=p) as ctx:
        for fl in ((4, 3, 2, 1) if not os.environ.get("HT_ONLY4") else (4,)):
            for pinned in (False, True):
                ctx.bench_callback_stream(recs, n, 32, 0, 4, 8, 16, fl, producer_pinned=pinned)
                sec, _ = ctx.bench_callback_stream(recs, n, 32, 0, 4, 8, reps, fl, producer_pinned=pinned)
                print("in flight %d pinned %d: %.1f sweeps/s" % (fl, pinned, reps / sec), flush=True)


if __name__ == "__main__":
    main()
