#!/usr/bin/env python
"""One conv geometry, one tile shape, N launches — the subject for PMC passes:
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY ... -- python tools/conv_one.py 128 128 11 1 39936 0 20
MI355TTS_BENCH_ABLATE in the environment selects an ablation mask (see tools/conv_probe.py)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd.engine import Engine  # noqa: E402

Cin, Cout, K, d, L, tile, iters = (int(v) for v in sys.argv[1:8])
eng = Engine(0)
ms = eng.bench_conv1d(1, Cin, Cout, K, d, L, tile, iters)
print(f"C{Cin}->{Cout} K{K} d{d} L{L} tile{tile}: {ms * 1e3:.1f} us  {2.0 * Cin * Cout * K * L / ms / 1e9:.1f} TF")
eng.close()
