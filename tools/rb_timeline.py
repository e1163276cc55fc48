#!/usr/bin/env python
"""Summarise a workgroup timeline dumped by tools/probe/rb_diag.hip: launch span, resident-workgroup curve,
per-member workgroup durations, per-CU busy spans, ramp and tail losses.  python tools/rb_timeline.py <dump> [slots]"""
import sys
from collections import defaultdict

path = sys.argv[1]
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
wg, chunks = [], []
for line in open(path):
    if line.startswith("#"):
        continue
    p = line.split()
    if p[0] == "C":
        chunks.append(tuple(int(v) for v in p[1:]))
    else:
        wg.append(tuple(int(v) for v in p))
span = max(w[3] for w in wg)
print(f"{len(wg)} workgroups, span {span / 100:.1f} us")
# resident workgroups over time
ev = sorted([(w[2], 1) for w in wg] + [(w[3], -1) for w in wg])
area, cur, last = 0, 0, 0
hist = defaultdict(int)
for t, d in ev:
    area += cur * (t - last)
    hist[cur * 16 // slots] += t - last  # sixteenths of the slots
    last, cur = t, cur + d
print(f"mean resident workgroups {area / span:.0f} of {slots} slots = {area / span / slots:.3f}")
print("time by occupancy (sixteenths of the slots): " + " ".join(f"{k}:{100 * v / span:.1f}%" for k, v in sorted(hist.items())))
for m in sorted({w[1] for w in wg}):
    d = sorted((w[3] - w[2]) / 100 for w in wg if w[1] == m)
    b = sorted(w[2] / 100 for w in wg if w[1] == m)
    e = sorted(w[3] / 100 for w in wg if w[1] == m)
    print(f"member {m}: {len(d)} wgs, duration us min/med/max {d[0]:.1f}/{d[len(d) // 2]:.1f}/{d[-1]:.1f}; begins {b[0]:.1f}..{b[-1]:.1f}, ends {e[0]:.1f}..{e[-1]:.1f}")
# per CU: key = (xcc, se, sh, cu) from HW_ID (gfx9 layout: cu_id [11:8], sh_id [12], se_id [15:13]) and XCC_ID [3:0]
cu = defaultdict(list)
for w in wg:
    hw, xcc = w[6], w[7] & 0xF
    cu[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)].append(w)
ends = sorted(max(w[3] for w in v) / 100 for v in cu.values())
firsts = sorted(min(w[2] for w in v) / 100 for v in cu.values())
n = [len(v) for v in cu.values()]
print(f"{len(cu)} CUs seen; workgroups per CU min/max {min(n)}/{max(n)}; first begin per CU {firsts[0]:.1f}..{firsts[-1]:.1f} us; last end per CU min/med/max {ends[0]:.1f}/{ends[len(ends) // 2]:.1f}/{ends[-1]:.1f} us")
xc = defaultdict(int)
for k, v in cu.items():
    xc[k[0]] += len(v)
print("workgroups per XCC:", dict(sorted(xc.items())))
if chunks:
    # per (sample, wave): stamps s0 (chunk top) s1 (MFMA loop done) s2 (LDS store done) s3 (barrier passed), s_memtime cycles
    import statistics as st
    mf, stw, bar, top = [], [], [], []
    byw = defaultdict(list)
    for s, w, c, s0, s1, s2, s3 in chunks:
        if s1 and s0:
            mf.append(s1 - s0)
        if s2 and s1:
            stw.append(s2 - s1)
        if s3 and s2:
            bar.append(s3 - s2)
        byw[(s, w)].append((c, s0, s3))
    for v in byw.values():
        v.sort()
        for (c0, a0, b0), (c1, a1, b1) in zip(v, v[1:]):
            if b0 and a1:
                top.append(a1 - b0)
    q = lambda x: f"{st.median(x):.0f} (p10 {sorted(x)[len(x) // 10]:.0f}, p90 {sorted(x)[9 * len(x) // 10]:.0f})" if x else "-"
    print(f"chunk phases, cycles: MFMA loop {q(mf)}; activation -> LDS {q(stw)}; barrier wait {q(bar)}; barrier -> next chunk top {q(top)}")
