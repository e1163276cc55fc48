#!/bin/bash
# whole-call coalescing (host_join.h) in the native fp16 mode: headline form at call_coalesce = 0 / 2 / 3 / 4, A B C D twice
O=gpurun_out/${1:-r06_f16_cc}
mkdir -p $O
for rep in 1 2; do
for cc in 0 2 3 4; do
  python bench.py --precision f16 --set-option call_coalesce=$cc --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch > $O/cc${cc}_$rep.json 2> $O/cc${cc}_$rep.err
  python - $O/cc${cc}_$rep.json $cc <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ss=d.get('steady_state',{})
print('call_coalesce', sys.argv[2], 'utt/s %.1f'%d['value'], 'steady', {k:round(v,1) for k,v in ss.items() if isinstance(v,(int,float))}, 'lat %.3f'%d['latency_ms_single_stream'])
PY
done
done
