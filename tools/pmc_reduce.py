#!/usr/bin/env python
"""Reduce a rocprofv3 counter_collection.csv to per-kernel sums/averages.
`pmc_reduce.py file.csv [--split-workgroups KERNEL_SUBSTRING N]`: dispatches of kernels whose name contains the substring
are reported in two rows, "<= N workgroups" and "> N workgroups" (rb_group_kernel runs both the 256-channel stage, ~470
workgroups at batch 1, and the 128-channel stage, ~1900: one kernel name, two launch geometries)."""
import collections
import csv
import re
import sys

split_sub, split_n = None, 0
if "--split-workgroups" in sys.argv:
    i = sys.argv.index("--split-workgroups")
    split_sub, split_n = sys.argv[i + 1], int(sys.argv[i + 2])
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("mi355tts::", "")
    if split_sub and split_sub in short:
        grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        wg = int(r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or 1)
        short += f" [<= {split_n} workgroups]" if grid // max(wg, 1) <= split_n else f" [> {split_n} workgroups]"
    key = (short, r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "sum", "avg_per_dispatch"])
for (k, c), (n, s) in agg.items():
    w.writerow([k, c, n, f"{s:.6g}", f"{s / n:.6g}"])
