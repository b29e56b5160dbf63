#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`
in ROCm 7.2) into the per-kernel table committed under profiles/.
    python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r1_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(workgroup_x), max(grid_x), max(grid_y) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-52s %6s %11s %10s %10s %10s %6s %5s %5s %6s %7s %5s %9s %7s" % (
        "kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%", "vgpr", "sgpr", "lds_B", "scratch", "wg_x", "grid_x", "grid_y"))
    for r in rows:
        print("%-52s %6d %11.3f %10.1f %10.1f %10.1f %6.1f %5d %5d %6d %7d %5d %9d %7d" % (
            r[0][:52], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
            r[6] + r[7], r[8], r[9], r[10], r[11], r[12], r[13]))


if __name__ == "__main__":
    main(sys.argv[1])
