#!/usr/bin/env python
"""Collapse rocprofv3 --pmc CSV output (counter_collection.csv files under a directory tree) into one
markdown table: per kernel (name prefix filter) the mean counter value per dispatch.
usage: pmc_summary.py <dir> [kernel-substring]"""
import csv
import os
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "prad::"
acc = defaultdict(lambda: defaultdict(list))
for dp, _, files in os.walk(root):
    for f in files:
        if f.endswith("counter_collection.csv"):
            with open(os.path.join(dp, f)) as fh:
                for row in csv.DictReader(fh):
                    k = row.get("Kernel_Name", "")
                    if flt not in k:
                        continue
                    short = k.split("(")[0].replace("void ", "")
                    acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print("### `%s`" % k)
    print("| counter | mean per dispatch | dispatches |")
    print("|---|---:|---:|")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("| %s | %.4g | %d |" % (c, sum(v) / len(v), len(v)))
    print()
