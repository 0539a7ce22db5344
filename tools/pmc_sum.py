#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per (kernel, counter): python tools/pmc_sum.py <dir> [name-filter]"""
import collections, csv, glob, sys
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:90]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        d[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in d.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f"   {name:32s} {v:14.4e}  ({n[(k, name)]} records)")
