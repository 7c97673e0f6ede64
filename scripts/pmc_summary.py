#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, launches and mean counter value per launch."""
import csv
import collections
import sys

path = sys.argv[1]
acc = collections.OrderedDict()
with open(path) as f:
    rd = csv.DictReader(f)
    for row in rd:
        name = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
        ctr = row.get("Counter_Name", "")
        val = float(row.get("Counter_Value", "0") or 0)
        did = row.get("Dispatch_Id", "")
        key = (name, ctr)
        d = acc.setdefault(key, {})
        d[did] = d.get(did, 0.0) + val
print("%-40s %-12s %8s %16s" % ("kernel", "counter", "launches", "mean_per_launch"))
for (name, ctr), d in acc.items():
    print("%-40s %-12s %8d %16.1f" % (name[:40], ctr, len(d), sum(d.values()) / len(d)))
