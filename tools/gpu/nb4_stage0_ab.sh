#!/bin/bash
# 128-column tiles also for the 256-channel stage (234 workgroups at 617 frames: fewer than CUs)?  MI355TTS_RB_NB4_MIN_TILES = 768 (default rule) vs 200.
out=gpurun_out/${1:-r06_nb4s0}
mkdir -p $out
for v in 768 200; do MI355TTS_RB_NB4_MIN_TILES=$v python tools/wave_hash.py high 617 2>&1 | tail -1 | cut -c1-400; done | tee $out/bits.txt
for i in 1 2; do
  for v in 768 200; do
    MI355TTS_RB_NB4_MIN_TILES=$v timeout 600 python bench.py --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch --no-half-mode \
      > $out/nb4_${v}_$i.json 2> $out/nb4_${v}_$i.err
    python - $out/nb4_${v}_$i.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["profile_ms_per_step"]
bk = d["roofline"]["by_kernel"].get("conv_mfma.hifigan_resblock", {})
print(f"min_tiles={sys.argv[2]} utt/s {d['value']:.1f} latency {d['latency_ms_single_stream']:.3f} ms resblock {p['conv_mfma.hifigan_resblock']:.3f} steady {(d.get('steady_state') or {}).get('utterances_per_sec')}",
      {k: round(v["avg_us"], 1) for k, v in bk.items()})
PY
  done
done 2>&1 | tee $out/summary.txt
