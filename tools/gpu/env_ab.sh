#!/bin/bash
# A B A B of one environment switch on the f32 headline: tools/gpu/env_ab.sh <out dir under gpurun_out> VAR valueA valueB [extra bench args]
out=gpurun_out/$1; var=$2; va=$3; vb=$4; shift 4
mkdir -p $out
for i in 1 2; do
  for v in $va $vb; do
    env $var=$v timeout 600 python bench.py --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch --no-half-mode "$@" \
      > $out/${var}_${v}_$i.json 2> $out/${var}_${v}_$i.err
    python - $out/${var}_${v}_$i.json "$var=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["profile_ms_per_step"]
bk = d["roofline"]["by_kernel"].get("conv_mfma.hifigan_resblock", {})
print(f"{sys.argv[2]} utt/s {d['value']:.1f} latency {d['latency_ms_single_stream']:.3f} ms resblock {p['conv_mfma.hifigan_resblock']:.3f} ups {p['conv_mfma.hifigan_upsample']:.3f} "
      f"steady {(d.get('steady_state') or {}).get('utterances_per_sec')} glow_under_load {d.get('glow_under_load_ms')}",
      {k: (v["launches"], round(v["avg_us"], 1)) for k, v in bk.items()})
PY
  done
done 2>&1 | tee $out/summary.txt
