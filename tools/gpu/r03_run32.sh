cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -8
B="python bench.py --no-cpu-baseline --no-config3 --no-config5"
run() {  # label, lib
  timeout 300 $B --library larynx_amd/$2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d.get('half_mode',{}); p=d['profile_ms_per_step']
print('$1', 'utt/s', round(d['value'],1), 'lat', round(d['latency_ms_single_stream'],3), 'frac', round(d['roofline']['frac'],4), 'ups', round(p.get('conv_mfma.hifigan_upsample',0),4), 'io', round(p.get('conv_mfma.hifigan_pre_post',0),4), 'half', round(h.get('utterances_per_sec',0),1), round(h.get('latency_ms_single_stream',0),3), 'c4', round(d['config4']['utterances_per_sec']), round(d['config4']['ms_per_call'],3), sorted(p.keys()) if '$1'=='base' else '')"
}
for i in 1 2; do
  run base libmi355tts_base.so
  run defer_cap libmi355tts.so
  run defer_uncap libmi355tts_uncap.so
done | tee $O/ab_ups_defer.log
