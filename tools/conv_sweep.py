#!/usr/bin/env python
"""Per-layer micro-benchmark of conv_mfma_kernel on the HiFi-GAN 'high' ResBlock
geometries of the standard utterance (F = 624), every tile shape.
Run on the GPU box:  python tools/conv_sweep.py [frames]"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd.engine import Engine  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 624
eng = Engine(0, library_path=os.environ.get("MI355TTS_LIB"))  # experiments: an alternative build
SHAPES = tuple(int(t) for t in os.environ.get("SWEEP_SHAPES", "3,0,1,2").split(","))
layers = []
for stage, (C, mul) in enumerate([(256, 8), (128, 64), (64, 128), (32, 256)]):
    for K in (3, 7, 11):
        for d in (1, 5):
            layers.append((f"s{stage} C{C} K{K} d{d}", C, C, K, d, F * mul))
layers += [("glow WN in 192->384 K5", 192, 384, 5, 1, F // 2), ("glow 1x1 192->384", 192, 384, 1, 1, F // 2),
           ("enc ffn1 192->768 K3", 192, 768, 3, 1, 120), ("conv_pre 80->512 K7", 80, 512, 7, 1, F)]
print(f"{'layer':28s} {'GFLOP':>7s} " + " ".join(f"{'tile'+str(t)+' us':>10s} {'TF':>6s}" for t in SHAPES) + "   auto")
for name, Cin, Cout, K, d, L in layers:
    gf = 2.0 * Cin * Cout * K * L / 1e9
    cells = []
    for t in SHAPES + (-1,):
        ms = eng.bench_conv1d(1, Cin, Cout, K, d, L, t, 20)
        cells.append((ms * 1e3, gf / ms))
    print(f"{name:28s} {gf:7.2f} " + " ".join(f"{us:10.1f} {tf:6.1f}" for us, tf in cells[:-1]) + f"   {cells[-1][0]:8.1f} {cells[-1][1]:6.1f}")
eng.close()
