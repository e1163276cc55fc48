#!/bin/bash
# wn_f16_kernel with its weights warm in L2: MI355TTS_WN_REPEAT = 1 / 2 / 3 launches per block inside ONE event pair (the launch is
# idempotent) — the step from 1 to 2 is what a launch costs when the block's 3.4 MB of fragments are L2 hits.
# Usage: tools/gpu/wn_f16_repeat.sh <out dir under gpurun_out>
out=gpurun_out/${1:-r06_wn_repeat}
mkdir -p $out
for r in 1 2 3 1 2; do
  MI355TTS_WN_REPEAT=$r timeout 600 python bench.py --precision f16 --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch --no-steady-state \
    > $out/repeat_$r.json 2> $out/repeat_$r.err
  python - $out/repeat_$r.json $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d["roofline"]["by_kernel"]["glow_top"]
for k, v in g.items():
    if "wn_f16" in k:
        print(f"repeat {sys.argv[2]}: {k} launches {v['launches']} avg {v['avg_us']:.1f} us per event pair; latency {d['latency_ms_single_stream']:.3f} ms; utt/s {d['value']:.1f}")
PY
done
