#!/bin/bash
# A/B of an option in the native fp16 mode: tools/gpu/f16_opt_ab.sh <tag> <option> [<bench args>]  -> option = 1 / 0, A B A B
T=$1; OPT=$2; shift; shift
O=gpurun_out/$T
mkdir -p $O
for rep in 1 2; do
for val in 1 0; do
  python bench.py --precision f16 --set-option $OPT=$val --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch "$@" > $O/${OPT}${val}_$rep.json 2> $O/${OPT}${val}_$rep.err
  python - $O/${OPT}${val}_$rep.json $OPT=$val <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p=d.get('profile_ms_per_step',{})
print(sys.argv[2], 'utt/s %.1f'%d['value'], 'lat %.3f ms'%d.get('latency_ms_single_stream',0), 'resblock %.3f ups %.3f prepost %.3f elem %.3f'%(p.get('conv_mfma.hifigan_resblock',0),p.get('conv_mfma.hifigan_upsample',0),p.get('conv_mfma.hifigan_pre_post',0),p.get('elementwise',0)), 'glow_under_load', d.get('glow_under_load_ms'))
bk=d['roofline'].get('by_kernel',{}).get('conv_mfma.hifigan_resblock',{})
print('   ', {k:round(v['avg_us'],1) for k,v in bk.items()})
PY
done
done
