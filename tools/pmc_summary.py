#!/usr/bin/env python3
"""Average one PMC counter per (kernel, grid size) from a rocprofv3 counter_collection.csv."""
import collections
import csv
import glob
import sys

pat, counter = sys.argv[1], sys.argv[2]
f = glob.glob(pat)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == counter and ("mi355" in r["Kernel_Name"]):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        key = (name.split("(")[0][:78], r["Grid_Size"])
        agg.setdefault(key, []).append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-80s grid %-10s n=%-4d avg %s = %.1f" % (k[0], k[1], len(v), counter, sum(v) / len(v)))
