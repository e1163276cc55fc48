#!/usr/bin/env python
"""Reduce a rocprofv3 counter_collection.csv to per-kernel sums/averages."""
import collections
import csv
import re
import sys

rows = csv.DictReader(open(sys.argv[1]))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("mi355tts::", "")
    key = (short, r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "sum", "avg_per_dispatch"])
for (k, c), (n, s) in agg.items():
    w.writerow([k, c, n, f"{s:.6g}", f"{s / n:.6g}"])
