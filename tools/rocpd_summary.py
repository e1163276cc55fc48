#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (`--kernel-trace --stats`) as a
per-kernel table: calls, total/avg/min/max duration, share of GPU time.
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.md"""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels").fetchall()
    agg = {}
    for name, s, e, gx, gy, gz, wx in rows:
        m = re.search(r"conv_mfma_kernel<([^>]*)>", name)
        short = f"conv_mfma_kernel<{m.group(1)}>" if m else re.sub(r"\(.*", "", name)
        short = short.replace("mi355tts::", "").replace("void ", "")
        a = agg.setdefault(short, [0, 0, 1e30, 0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 kernel-trace summary of `{path}`\n")
    print(f"total kernel time {tot/1e3:.3f} ms over {sum(a[0] for a in agg.values())} dispatches\n")
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1]:.1f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100*a[1]/tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
