#!/usr/bin/env python3
"""Start / end of the kernels of one step from a rocprofv3 kernel-trace database: does k_front_finish's first part run next to the sort?
    python tools/r6_overlap.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall() if "stream_id" in [r[1] for r in db.execute("pragma table_info(kernels)")] else \
       [(n, s, e, 0) for n, s, e in db.execute("select name, start, end from kernels order by start")]
# last occurrence of k_front as the anchor of a step without the event brackets
idx = [i for i, r in enumerate(rows) if r[0].startswith("k_front(")]
i0 = idx[len(idx) // 2]
t0 = rows[i0][1]
for n, s, e, q in rows[i0 - 1:i0 + 24]:
    print("%-46s start %9.1f us  end %9.1f us  dur %8.1f  stream %s" % (n[:46], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q))
