"""Timeline of a rocprofv3 --kernel-trace run (the rocpd database it writes, <out>_results.db, or a kernel-trace CSV):
per kernel the mean duration, the gap to the previous kernel of the same queue and how many kernels of OTHER queues were
running when it started; per queue / stream the number of dispatches.  Used on the callback path (tools/host_times.py under
rocprofv3): two of the four slots' streams turned out to share a hardware queue.
    cd /tmp && rocprofv3 --kernel-trace -d out -o t -- python $REPO/tools/host_times.py 200
    python tools/trace_overlap.py out/t_results.db [first_row]"""
import csv
import sqlite3
import sys
from collections import Counter, defaultdict


def load(path):
    if path.endswith(".db"):
        con = sqlite3.connect(path)
        return [(r[0], r[1], r[2].split("(")[0], "q%s/s%s" % (r[3], r[4]))
                for r in con.execute("select start, end, name, queue_id, stream_id from kernels order by start")]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], "q" + r.get("Queue_Id", "0")))
    return sorted(rows)


def main():
    rows = load(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2   # the second half: steady state
    rows = rows[skip:]
    print("dispatches per queue/stream:", dict(Counter(r[3] for r in rows)))
    byq = defaultdict(list)
    for r in rows:
        byq[r[3].split("/")[0]].append(r)
    dur, gap, conc = defaultdict(list), defaultdict(list), defaultdict(list)
    for v in byq.values():
        for i, r in enumerate(v):
            dur[r[2]].append(r[1] - r[0])
            if i:
                gap[r[2]].append(r[0] - v[i - 1][1])
    for i, r in enumerate(rows):
        q = r[3].split("/")[0]
        conc[r[2]].append(sum(1 for o in rows[max(0, i - 40):i] if o[3].split("/")[0] != q and o[1] > r[0]))
    print("%-28s %6s %9s %9s %6s" % ("kernel", "n", "dur us", "gap us", "others"))
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        print("%-28s %6d %9.2f %9.2f %6.2f" % (k[:28], len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e3,
                                             sum(gap[k]) / max(1, len(gap[k])) / 1e3, sum(conc[k]) / len(conc[k])))
    span = rows[-1][1] - rows[0][0]
    busy = sum(r[1] - r[0] for r in rows)
    print("span %.1f us, kernel time %.1f us: %.2f kernels running on average" % (span / 1e3, busy / 1e3, busy / span))


if __name__ == "__main__":
    main()
