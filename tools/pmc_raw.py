#!/usr/bin/env python3
"""Per kernel, the mean of every counter in a rocprofv3 PMC pass (raw values; no corrections).
  usage: tools/pmc_raw.py <dir> [substring of the kernel name]"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
            continue
        agg[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:28s} n={len(v):4d} mean {sum(v) / len(v):.4g}")
