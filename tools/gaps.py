"""Time line of one lnprob call from a rocprofv3 kernel trace: each kernel's duration and the gap to the one before it.
    python tools/gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                                  # the timed half of the run: clocks settled
dur, gap = defaultdict(list), defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None and s - prev_end < 200000:
        gap[name].append(s - prev_end)
    prev_end = e
for k in dur:
    d, g = sorted(dur[k]), sorted(gap[k]) or [0]
    print("%-62s n %5d  dur median %7.2f us   gap before: median %6.2f us  p10 %6.2f" % (
        k, len(d), d[len(d) // 2] / 1e3, g[len(g) // 2] / 1e3, g[len(g) // 10] / 1e3))
