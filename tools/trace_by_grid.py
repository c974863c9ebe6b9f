#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid, LDS): calls, average and minimum duration.  stats.csv averages a
kernel over every shape it ran at; this splits them so a launch in the pipeline can be compared with the same shape in
tools/kernel_bench.py.      usage: tools/trace_by_grid.py <dir with *_kernel_trace.csv> [min total ms] [list:<substring>]
list:<substring> also prints every launch of the kernels whose name contains it, in launch order: start (ms), us, workgroups."""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
listed = [a[5:] for a in sys.argv[3:] if a.startswith("list:")]
acc = collections.defaultdict(list)
seq = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for cut in ("void ", "svr2::", "(anonymous namespace)::"):
            name = name.replace(cut, "")
        wgs = 1
        for ax in "XYZ":
            wgs *= max(int(r[f"Grid_Size_{ax}"]) // max(int(r[f"Workgroup_Size_{ax}"]), 1), 1)
        key = (name[:70], wgs, r.get("LDS_Block_Size", "?"))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if any(sub in name for sub in listed):
            seq.append((int(r["Start_Timestamp"]), acc[key][-1], key[0][:40], key[1]))
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
print(f"{'kernel':70s} {'wgs':>7s} {'lds':>7s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'total_ms':>9s}")
for (name, wgs, lds), d in rows:
    if sum(d) / 1e3 < floor:
        continue
    print(f"{name:70s} {wgs:7d} {lds:>7s} {len(d):6d} {sum(d)/len(d):9.1f} {min(d):9.1f} {sum(d)/1e3:9.2f}")
if seq:
    seq.sort()
    t0 = seq[0][0]
    for st, us, name, wgs in seq:
        print(f"{(st - t0) / 1e6:10.1f} ms  {us:10.1f} us  {wgs:7d}  {name}")
