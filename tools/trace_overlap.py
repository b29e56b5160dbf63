"""Timeline of a rocprofv3 --kernel-trace CSV: per-kernel mean duration, gap to the previous kernel of the same queue,
and how many kernels of OTHER queues were running at a kernel's start (overlap between the callback path's slots)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2   # look at the second half: steady state
rows = rows[skip:]
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
print("queues:", {q: len(v) for q, v in byq.items()})
dur = defaultdict(list)
gap = defaultdict(list)
for q, v in byq.items():
    for i, r in enumerate(v):
        dur[r[2]].append(r[1] - r[0])
        if i:
            gap[r[2]].append(r[0] - v[i - 1][1])
conc = defaultdict(list)
for i, r in enumerate(rows):
    n = sum(1 for o in rows[max(0, i - 40):i] if o[3] != r[3] and o[1] > r[0])
    conc[r[2]].append(n)
print("%-28s %6s %9s %9s %6s" % ("kernel", "n", "dur us", "gap us", "conc"))
tot_d = tot_g = 0.0
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = sum(dur[k]) / len(dur[k]) / 1e3
    g = sum(gap[k]) / max(1, len(gap[k])) / 1e3
    tot_d += d
    tot_g += g
    print("%-28s %6d %9.2f %9.2f %6.2f" % (k[:28], len(dur[k]), d, g, sum(conc[k]) / len(conc[k])))
print("sum of mean durations %.1f us, of mean gaps %.1f us" % (tot_d, tot_g))
span = rows[-1][1] - rows[0][0]
busy = sum(r[1] - r[0] for r in rows)
print("span %.1f us, kernel time %.1f us: %.2f kernels running on average" % (span / 1e3, busy / 1e3, busy / span))
