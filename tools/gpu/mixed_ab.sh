#!/bin/bash
# probe: the exact f32 vocoder behind the fp16 acoustic mode (bench.py --acoustic-f16) against the f32 headline, A B A B
out=gpurun_out/${1:-r06_mixed}
mkdir -p $out
for i in 1 2; do
  for v in "" "--acoustic-f16"; do
    n=$([ -z "$v" ] && echo f32 || echo mixed)
    timeout 600 python bench.py --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch --no-half-mode $v > $out/${n}_$i.json 2> $out/${n}_$i.err
    python - $out/${n}_$i.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["profile_ms_per_step"]
print(f"{sys.argv[2]:6s} utt/s {d['value']:.1f} latency {d['latency_ms_single_stream']:.3f} ms glow_dec {p['conv_mfma.glow_decoder']:.3f} steady {(d.get('steady_state') or {}).get('utterances_per_sec'):.1f} "
      f"glow_under_load {d.get('glow_under_load_ms'):.3f} steady {(d.get('steady_state') or {}).get('glow_under_load_ms'):.3f}")
PY
  done
done 2>&1 | tee $out/summary.txt
