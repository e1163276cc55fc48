#!/usr/bin/env python
"""A handful of conv micro-benchmark launches for PMC collection:
rocprofv3 --pmc ... -- python tools/conv_probe.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd.engine import Engine  # noqa: E402

eng = Engine(0)
cases = [(128, 128, 11, 1, 39936, 0), (128, 128, 11, 1, 39936, 1), (128, 128, 3, 1, 39936, 0), (128, 128, 3, 1, 39936, 1),
         (32, 32, 3, 1, 159744, 1), (256, 256, 11, 1, 4992, 0), (128, 128, 11, 1, 399360, 2), (128, 128, 11, 1, 399360, 1)]
for Cin, Cout, K, d, L, t in cases:
    ms = eng.bench_conv1d(1, Cin, Cout, K, d, L, t, 5)
    print(Cin, Cout, K, d, L, "tile", t, f"{ms*1e3:.1f} us", f"{2.0*Cin*Cout*K*L/ms/1e9:.1f} TF")
eng.close()
