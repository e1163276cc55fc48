#!/bin/bash
# A/B of builds / environment knobs on one GPU box: single-stream latency and the profiled class times (ms per utterance).
# Usage: tools/lat_sweep.sh "VAR=a" "MI355TTS_LIB=larynx_amd/lib_x.so" ...
for kv in "$@"; do
  env $kv python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --concurrency 1 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['profile_ms_per_step']
print('$kv', 'lat_ms', round(d['latency_ms_single_stream'],3), ' '.join(f'{k.split(\".\")[-1]}={v:.3f}' for k,v in p.items() if v > 0.02))
"
done
