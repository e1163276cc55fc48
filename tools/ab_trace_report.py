#!/usr/bin/env python
"""Compare the per-kernel average durations of the two runs tools/ab_trace.sh left under
gpurun_out/ab0 and gpurun_out/ab1 (first argument of the script = ab0)."""
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


a, b = load(0), load(1)
ta = tb = 0.0
for n, (c, us) in sorted(a.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:22]:
    if n in b:
        print(f"{n:55s} {c:5d} {us:8.1f} -> {b[n][1]:8.1f}  {100 * (b[n][1] / us - 1):+.1f}%")
for n, (c, us) in a.items():
    if n in b:
        ta += c * us
        tb += b[n][0] * b[n][1]
print(f"total kernel time: {ta / 1e3:.2f} ms -> {tb / 1e3:.2f} ms ({100 * (tb / ta - 1):+.2f}%)")
