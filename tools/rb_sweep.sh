#!/bin/bash
# A/B sweep of environment knobs on one GPU box: prints the ResBlock class time (ms per utterance, profiled
# pass of bench.py) and the single-stream latency per setting.  Usage: tools/rb_sweep.sh "VAR=a VAR=b ..."
for kv in "$@"; do
  env $kv python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --concurrency 1 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$kv', 'class_ms', round(d['profile_ms_per_step']['conv_mfma.hifigan_resblock'],3), 'frac', round(r['frac'],4), 'lat_ms', round(d['latency_ms_single_stream'],3), 'ups', round(d['profile_ms_per_step']['conv_mfma.hifigan_upsample'],3))
"
done
