#!/bin/bash
# A/B of library builds in the native fp16 mode: tools/gpu/f16_ab.sh <tag> <lib> [<lib> ...]   (libs = paths of libmi355tts builds)
# each build twice, interleaved (A B A B): headline form (8 calls in flight) with the per-class profile
T=$1; shift
O=gpurun_out/$T
mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  n=$(basename $lib .so)
  python bench.py --precision f16 --library $lib --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch ${F16_AB_EXTRA:-} > $O/${n}_$rep.json 2> $O/${n}_$rep.err
  python - $O/${n}_$rep.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d.get('profile_ms_per_step',{})
print(sys.argv[2], 'utt/s %.1f'%d['value'], 'lat %.3f ms'%d.get('latency_ms_single_stream',0), 'resblock %.3f ups %.3f prepost %.3f elem %.3f'%(p.get('conv_mfma.hifigan_resblock',0),p.get('conv_mfma.hifigan_upsample',0),p.get('conv_mfma.hifigan_pre_post',0),p.get('elementwise',0)), 'glow_under_load', d.get('glow_under_load_ms'))
PY
done
done
