#!/usr/bin/env python
"""gpurun_out/<dir>/ (tools/gpu/r06_final.sh: a --kernel-trace --stats pass and two SQ passes of `bench.py --precision f16`, one call in
flight) -> profiles/<round>_f16_kernel_stats.csv, _f16_pmc_by_kernel.csv, _f16_summary.md.   Usage: python tools/f16_summary.py r06_f16"""
import csv
import shutil
import sys
from pathlib import Path

D = sys.argv[1] if len(sys.argv) > 1 else "r06_f16"
SRC = Path("gpurun_out") / D
DST = Path("profiles")
R = D.split("_")[0]


def load(path):
    d = {}
    for r in csv.DictReader(open(path)):
        d.setdefault(r["kernel"], {})[r["counter"]] = (int(r["dispatches"]), float(r["sum"]))
    return d


shutil.copy(SRC / "trace" / "trace_kernel_stats.csv", DST / f"{R}_f16_kernel_stats.csv")
rows = []
for name in ("sq", "sq2"):
    rows += list(csv.reader(open(SRC / f"pmc_{name}" / f"{name}_counter_collection_by_kernel.csv")))[0 if name == "sq" else 1:]
with open(DST / f"{R}_f16_pmc_by_kernel.csv", "w", newline="") as f:
    csv.writer(f).writerows(rows)
pmc = load(DST / f"{R}_f16_pmc_by_kernel.csv")
stats = {}
for r in csv.DictReader(open(DST / f"{R}_f16_kernel_stats.csv")):
    stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))


def short(n):
    return n.replace("void mi355tts::", "").split("(")[0]


out = [f"# {R} fp16 mode: rocprofv3 of `python bench.py --precision f16 --steps 6 --warmup 2 --concurrency 1 ...` (tools/gpu/r06_final.sh: a --kernel-trace --stats pass + two SQ passes)",
       "",
       "One call at a time on one stream (the single-stream profile; under load the launches of 8 calls overlap).  Both models in fp16: the vocoder (conv_f16.h / pair_f16.h) and the",
       "WaveNets of GlowTTS' decoder (`wn_f16_kernel`, one launch per coupling block).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs);",
       "waves/SIMD = 4 x SQ_WAVE_CYCLES / the same denominator (average resident waves); VALU / LDS / VMEM = instructions per MFMA instruction.",
       "",
       "| kernel | dispatches | avg us (trace) | % time | MFMA busy | waves/SIMD | wait_any | wait_inst | VALU/MFMA | LDS/MFMA | VMEM rd/MFMA |",
       "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
by_short = {short(k): v for k, v in stats.items()}
for k, c in sorted(pmc.items(), key=lambda kv: -by_short.get(kv[0].split(" [")[0], (0, 0, 0))[2]):
    base = k.split(" [")[0]
    st = by_short.get(base)
    if st is None or "GRBM_GUI_ACTIVE" not in c:
        continue
    gui = c["GRBM_GUI_ACTIVE"][1] / 8 * 1024
    g = lambda n: c.get(n, (0, 0.0))[1]
    mf = g("SQ_INSTS_MFMA")
    per = lambda n: (g(n) / mf) if mf > 0 else float("nan")
    wc = g("SQ_WAVE_CYCLES")
    out.append(f"| `{k}` | {c['GRBM_GUI_ACTIVE'][0]} | {st[1]:.1f} | {st[2]:.2f} | {g('SQ_VALU_MFMA_BUSY_CYCLES') / gui:.3f} | {4 * wc / gui:.2f} | "
               f"{(g('SQ_WAIT_ANY') / wc) if wc else 0:.2f} | {(g('SQ_WAIT_INST_ANY') / wc) if wc else 0:.2f} | {per('SQ_INSTS_VALU'):.1f} | {per('SQ_INSTS_LDS'):.2f} | {per('SQ_INSTS_VMEM_RD'):.2f} |")
(DST / f"{R}_f16_summary.md").write_text("\n".join(out) + "\n")
print("\n".join(out[6:26]))
