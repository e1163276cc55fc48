#!/usr/bin/env python
"""Turn gpurun_out/<round>/ (tools/profile_round.sh) into the committed evidence
under profiles/: kernel stats, PMC-derived HBM traffic and MFMA utilisation."""
import csv
import json
import shutil
import sys
from pathlib import Path

R = sys.argv[1] if len(sys.argv) > 1 else "r02"
# optional second argument "medium": the passes over tools/config4_probe.py (BASELINE config 4: thorsten + 'medium', batch 8)
TAG = (sys.argv[2] + "_") if len(sys.argv) > 2 else ""
SRC = Path("gpurun_out") / R
DST = Path("profiles")
DST.mkdir(exist_ok=True)
ROUND = R
R = R + ("_" + TAG[:-1] if TAG else "")


def load(path):
    d = {}
    for r in csv.DictReader(open(path)):
        d.setdefault(r["kernel"], {})[r["counter"]] = (int(r["dispatches"]), float(r["sum"]))
    return d


shutil.copy(SRC / f"{TAG}trace" / "trace_kernel_stats.csv", DST / f"{R}_kernel_stats.csv")
for name in ("fetch", "write", "sq"):
    shutil.copy(SRC / f"{TAG}pmc_{name}" / f"{name}_counter_collection_by_kernel.csv", DST / f"{R}_pmc_{name}_by_kernel.csv")
fetch, write, sq = (load(DST / f"{R}_pmc_{n}_by_kernel.csv") for n in ("fetch", "write", "sq"))
stats = {r["Name"]: r for r in csv.DictReader(open(DST / f"{R}_kernel_stats.csv"))}


def short(n):
    import re

    return re.sub(r"\(.*", "", n).replace("void ", "").replace("mi355tts::", "")


rows = []
for full, st in stats.items():
    k = short(full)
    calls = int(st["Calls"])
    avg_us = float(st["AverageNs"]) / 1e3
    f = fetch.get(k, {}).get("FETCH_SIZE")
    w = write.get(k, {}).get("WRITE_SIZE")
    s = sq.get(k, {})
    util = None
    clk = None
    if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s and s["GRBM_GUI_ACTIVE"][1] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
        util = s["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (s["GRBM_GUI_ACTIVE"][1] / 8.0 * 1024.0)
    rows.append(dict(kernel=k, calls=calls, avg_us=avg_us, pct=float(st["Percentage"]),
                     fetch_kb=(f[1] / f[0]) if f else None, write_kb=(w[1] / w[0]) if w else None, mfma_util=util))
rows.sort(key=lambda r: -r["pct"])
with open(DST / f"{R}_summary.md", "w") as out:
    if TAG:
        out.write(f"# {R}: rocprofv3 summary of `python tools/config4_probe.py 10` — BASELINE config 4: de-de thorsten GlowTTS + hifi_gan 'medium', "
                  "ONE padded batch of 8 rows (P = 19 ... 120, 2200 frames) per fused call, single stream\n\n"
                  "(`mrf_small_kernel` = a whole 16- / 8-channel stage per launch; `pair_group_kernel` = the 64- / 32-channel stages)\n\n")
    else:
        out.write(f"# {R}: rocprofv3 summary of `python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1`\n\n"
                  "(the product schedule of a call that has the GPU to itself: one stream, the three MRF chains' same-geometry convs / fused pairs as ONE "
                  "grouped launch — `rb_group_kernel` (256- and 128-channel stages: round 4's continuous-stream tile; `conv_group_kernel` = the k-split tile for the utterance lengths the promotion rule leaves alone), `rb_pair_group_kernel` for the 64/32-channel stages)\n\n")
    out.write("Sources: `--kernel-trace --stats` (durations), separate `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and SQ passes "
              "(tools/profile_round.sh).  FETCH/WRITE are KB per dispatch as rocprofv3 reports them; on gfx950 FETCH_SIZE "
              "under-reports wide (16 B/lane) streaming reads by 2x (MI355X_MICROARCH.md §HBM) — the conv kernel reads its "
              "activations 4 B/lane and its weights 16 B/lane, so the read figure is a lower bound.  MFMA util = "
              "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs).\n\n")
    out.write("| kernel | calls | avg us | % time | FETCH KB | WRITE KB | MFMA util |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for r in rows:
        fk = f"{r['fetch_kb']:.0f}" if r["fetch_kb"] is not None else "-"
        wk = f"{r['write_kb']:.0f}" if r["write_kb"] is not None else "-"
        mu = f"{100*r['mfma_util']:.1f}%" if r["mfma_util"] is not None else "-"
        out.write(f"| `{r['kernel']}` | {r['calls']} | {r['avg_us']:.1f} | {r['pct']:.2f} | {fk} | {wk} | {mu} |\n")
# dominant kernel class = the HiFi-GAN ResBlock launches: LINEAR conv instances with K in {3,7,11}
# (wide stages) plus the fused conv-pair kernel (32/64-channel stages)
def is_dom(k):
    if TAG:
        return k.startswith("mrf_small_kernel")
    return k.startswith(("pair_group_kernel", "conv_group_kernel", "resblock_pair_kernel", "rb_group_kernel", "rb_pair_group_kernel", "rb_pair_kernel"))


dom = [r for r in rows if is_dom(r["kernel"])]
n = sum(r["calls"] for r in dom)
traffic = sum(r["calls"] * ((r["fetch_kb"] or 0) + (r["write_kb"] or 0)) for r in dom) * 1024.0 / n
# MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE counts a wide coalesced streaming read (16 B/lane: how these kernels
# read both activations and weights) at exactly half its bytes — double it before comparing with a byte count
traffic_corr = sum(r["calls"] * (2.0 * (r["fetch_kb"] or 0) + (r["write_kb"] or 0)) for r in dom) * 1024.0 / n
avg_us = sum(r["calls"] * r["avg_us"] for r in dom) / n
json.dump({"round": R, "kernel_class": ("mrf_small_kernel (the 16- and 8-channel stages of 'medium', one launch per stage)" if TAG else
                                        "HiFi-GAN ResBlock launches: rb_group_kernel / conv_group_kernel (256/128-channel stages) + rb_pair_group_kernel / pair_group_kernel (64/32-channel stages)"),
           "dispatches": n, "avg_us": avg_us, "hbm_bytes_per_launch_raw": traffic, "hbm_bytes_per_launch": traffic_corr,
           "note": "FETCH_SIZE / WRITE_SIZE (KB x 1024) per dispatch from separate PMC passes; `hbm_bytes_per_launch` applies the "
                   "guide's gfx950 correction (FETCH_SIZE counts 16-B/lane streaming reads at half their bytes: x2 on the read "
                   "side), `..._raw` is the uncorrected sum.  Infinity-Cache hits are counted, so this is memory-side traffic "
                   "of the L2s, not DRAM traffic"},
          open(DST / f"{R}_roofline_traffic.json", "w"), indent=1)
print(open(DST / f"{R}_summary.md").read())
print(open(DST / f"{R}_roofline_traffic.json").read())
