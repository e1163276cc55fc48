cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -8
B="python bench.py --no-cpu-baseline --no-config5"
run() {  # label, env
  env $2 timeout 400 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d.get('half_mode',{}); p=d['profile_ms_per_step']; c=d.get('glow_coalescing') or {}
print('$1', 'utt/s', round(d['value'],1), 'off', round(c.get('utterances_per_sec_with_the_option_off',0),1), 'rows/pass', round(c.get('rows_per_pass_in_the_timed_regions',0),2), 'lat', round(d['latency_ms_single_stream'],3), 'frac', round(d['roofline']['frac'],4), 'half', round(h.get('utterances_per_sec',0),1), round(h.get('latency_ms_single_stream',0),3), 'c3', round(d['config3']['utterances_per_sec'],1) if d.get('config3') else None, 'c4', round(d['config4']['utterances_per_sec']), round(d['config4']['ms_per_call'],3))"
}
for i in 1 2; do
  run coalesce X=1
  run off MI355TTS_NO_GLOW_COALESCE=1
done | tee $O/ab_coalesce.log
