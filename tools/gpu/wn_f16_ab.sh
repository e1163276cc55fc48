#!/bin/bash
# fp16 mode, acoustic model f32 / f16 (the decoder's WaveNets: wn_f16.h), A B A B inside one gpurun call + the device tests of the mode.
# Usage (from the repo root on the GPU box): tools/gpu/wn_f16_ab.sh <out dir under gpurun_out>
out=gpurun_out/${1:-r06_wn}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -k "f16 or half or full_size" -q -s > $out/tests.log 2>&1
echo "tests rc $?" >> $out/tests.log
for i in 1 2; do
  for a in f32 f16; do
    timeout 600 python bench.py --precision f16 --half-acoustic $a --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch \
      > $out/acoustic_${a}_$i.json 2> $out/acoustic_${a}_$i.err
  done
done
python - "$out" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/acoustic_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
        continue
    p = d.get("profile_ms_per_step", {})
    print(f"{os.path.basename(f):28s} utt/s {d['value']:7.1f}  latency {d.get('latency_ms_single_stream', 0):.3f} ms  glow_dec {p.get('conv_mfma.glow_decoder', 0):.3f}  "
          f"glow_enc {p.get('conv_mfma.glow_encoder', 0):.3f}  resblock {p.get('conv_mfma.hifigan_resblock', 0):.3f}  glow_under_load {d.get('glow_under_load_ms')}  "
          f"steady {(d.get('steady_state') or {}).get('utterances_per_sec')}")
PY
tail -5 $out/tests.log
