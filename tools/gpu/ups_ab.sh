#!/bin/bash
# the f32 upsamplers: compile-time MRF planes for the 64-row tile (MI355TTS_NO_UPS_PLANES = 1 / 0) and the column blocks per wave of the
# 128-row MRF tile (MI355TTS_UPS_NB = 2 / 1 / 4): bits, per-kernel times (single stream) and the headline
out=gpurun_out/${1:-r06_ups}
mkdir -p $out
for kv in "MI355TTS_NO_UPS_PLANES=1" "MI355TTS_NO_UPS_PLANES=0" "MI355TTS_UPS_NB=1" "MI355TTS_UPS_NB=4"; do echo -n "$kv "; env $kv python tools/wave_hash.py high 617 2>&1 | tail -1 | cut -c1-60; done | tee $out/bits.txt
for i in 1 2; do
  for kv in "MI355TTS_NO_UPS_PLANES=1" "MI355TTS_NO_UPS_PLANES=0" "MI355TTS_UPS_NB=1" "MI355TTS_UPS_NB=4"; do
    env $kv timeout 600 python bench.py --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch --no-half-mode \
      > $out/${kv}_$i.json 2> $out/${kv}_$i.err
    python - $out/${kv}_$i.json "$kv" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["profile_ms_per_step"]
bk = d["roofline"]["by_kernel"].get("conv_mfma.hifigan_upsample", {})
print(f"{sys.argv[2]} utt/s {d['value']:.1f} latency {d['latency_ms_single_stream']:.3f} ms ups {p['conv_mfma.hifigan_upsample']:.3f} steady {(d.get('steady_state') or {}).get('utterances_per_sec'):.1f}",
      {k: round(v["avg_us"], 1) for k, v in bk.items()})
PY
  done
done 2>&1 | tee $out/summary.txt
