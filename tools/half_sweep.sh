#!/bin/bash
# A/B of environment knobs for the bf16x3 mode on one GPU box (class time, latency).  Usage: tools/half_sweep.sh "VAR=a" "VAR=b" ...
for kv in "$@"; do
  env $kv python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-config3 --no-config5 --concurrency 1 --repeats 3 --precision bf16x3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$kv', 'class_ms', round(d['profile_ms_per_step']['conv_mfma.hifigan_resblock'],3), 'lat_ms', round(d['latency_ms_single_stream'],3))
"
done
