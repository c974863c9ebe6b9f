#!/usr/bin/env python3
"""Print what surrounds the largest GPU-idle gap of the last step in a rocprofv3 trace (kernel + memory-copy CSVs)."""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"M {r.get('Direction', '?')} {r.get('Bytes', r.get('Size', '?'))} B"))
ev.sort()
t_end = max(e[1] for e in ev)
tail = [e for e in ev if e[0] > t_end - 30_000_000]      # the last 30 ms
best, cur_e = (0, 0), tail[0][1]
for i, (s, e, n) in enumerate(tail[1:], 1):
    if s - cur_e > best[0]:
        best = (s - cur_e, i)
    cur_e = max(cur_e, e)
if len(sys.argv) > 2:      # context of every event whose name contains argv[2] instead of the largest gap
    for j, (s, e, n) in enumerate(tail):
        if sys.argv[2] in n:
            t0 = tail[max(0, j - 12)][0]
            print(f"-- context of event {j}")
            for s2, e2, n2 in tail[max(0, j - 12): j + 5]:
                print(f"  +{(s2 - t0) / 1e3:9.1f} us  dur {(e2 - s2) / 1e3:8.1f} us  {n2}")
    sys.exit(0)
g, i = best
print(f"largest gap in the last 30 ms: {g / 1e3:.1f} us before event {i}")
t0 = tail[max(0, i - 8)][0]
for s, e, n in tail[max(0, i - 8): i + 8]:
    print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  {n}")
