cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -8
B="python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode"
run() {  # label, lib
  timeout 400 $B --library larynx_amd/$2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['profile_ms_per_step']; c=d['config4']
print('$1', 'utt/s', round(d['value'],1), 'lat', round(d['latency_ms_single_stream'],3), 'ups', round(p.get('conv_mfma.hifigan_upsample',0),4), 'c4', round(c['utterances_per_sec']), round(c['ms_per_call'],3))"
}
for i in 1 2 3; do
  run base libmi355tts_base.so
  run defer libmi355tts.so
done | tee $O/ab_defer_m128.log
