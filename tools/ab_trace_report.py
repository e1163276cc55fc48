#!/usr/bin/env python
"""Compare per-kernel average durations of the A B A B runs tools/ab_trace.sh left under
gpurun_out/ab0..ab3 (A = first library argument: runs 0 and 2; B: runs 1 and 3)."""
import csv
import glob
import re


def load(i):
    f = glob.glob(f"gpurun_out/ab{i}/**/*kernel_stats.csv", recursive=True)[0]
    d = {}
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Name"].replace("void mi355tts::", ""))
        d[n] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    return d


runs = [load(i) for i in range(4)]


def avg(name, idx):
    vals = [runs[i][name][1] for i in idx if name in runs[i]]
    return sum(vals) / len(vals) if vals else None


ta = tb = 0.0
names = sorted(runs[0], key=lambda n: -runs[0][n][0] * runs[0][n][1])
for n in names:
    a, b = avg(n, (0, 2)), avg(n, (1, 3))
    if a is None or b is None:
        continue
    c = runs[0][n][0]
    ta += c * a
    tb += c * b
    if names.index(n) < 22:
        print(f"{n:55s} {c:5d} {a:8.1f} -> {b:8.1f}  {100 * (b / a - 1):+.1f}%   (A runs {runs[0][n][1]:.1f}/{runs[2][n][1]:.1f}, B runs {runs[1][n][1]:.1f}/{runs[3][n][1]:.1f})")
print(f"total kernel time: {ta / 1e3:.2f} ms -> {tb / 1e3:.2f} ms ({100 * (tb / ta - 1):+.2f}%)")
