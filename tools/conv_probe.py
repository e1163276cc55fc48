#!/usr/bin/env python
"""Conv micro-benchmark probe: a few geometries x tile shapes x ablation masks
(MI355TTS_BENCH_ABLATE: 1 = no activation staging, 2 = no weight loads, 4 = no barrier).
The product library ignores the masks; build a probe library first and point MI355TTS_LIB at it:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMI355TTS_ABLATION \
        larynx_amd/csrc/mi355tts.hip -o larynx_amd/lib_probe.so
(the runtime tests the probe build adds around the weight loads cost it ~10 % on their own:
compare its mask-0 column with the product library's).
rocprofv3 --pmc ... -- python tools/conv_probe.py"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd.engine import Engine  # noqa: E402

eng = Engine(0, library_path=os.environ.get("MI355TTS_LIB"))  # experiments: an alternative build
cases = [(128, 128, 11, 1, 39936, 0), (128, 128, 11, 1, 399360, 1), (128, 128, 11, 1, 399360, 2),
         (128, 128, 3, 1, 399360, 1), (32, 32, 3, 1, 1597440, 1), (256, 256, 11, 1, 4992, 3)]
masks = [int(m) for m in os.environ.get("PROBE_MASKS", "0,1,2,4,7").split(",")]
for Cin, Cout, K, d, L, t in cases:
    row = []
    for m in masks:
        os.environ["MI355TTS_BENCH_ABLATE"] = str(m)
        ms = eng.bench_conv1d(1, Cin, Cout, K, d, L, t, 5)
        row.append(f"ablate={m}: {ms*1e3:8.1f} us {2.0*Cin*Cout*K*L/ms/1e9:6.1f} TF")
    print(f"C{Cin} K{K} L{L} tile{t} | " + " | ".join(row))
eng.close()
