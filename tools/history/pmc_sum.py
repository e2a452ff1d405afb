#!/usr/bin/env python3
"""Sum a rocprofv3 --pmc counter per kernel name: python tools/pmc_sum.py <dir> <COUNTER> -> kernel, launches, mean value."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == sys.argv[2]:
            a = acc[r["Kernel_Name"][:70]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%-72s launches %5d  mean %14.1f" % (k, n, v / n))
