#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output directories: per kernel name, the mean of every counter over its dispatches.
usage: pmc_table.py <dir with g*/.../*counter_collection.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            k = k.split("(")[0][:48]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("== %s" % k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-44s n=%-3d mean %.6g" % (c, len(v), sum(v) / len(v)))
