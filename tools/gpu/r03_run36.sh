cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -8
B="python bench.py --no-cpu-baseline --no-config3 --no-config5"
run() {  # label, env
  env $2 timeout 400 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d.get('half_mode',{}); p=d['profile_ms_per_step']; c=d['config4']
print('$1', 'utt/s', round(d['value'],1), 'lat', round(d['latency_ms_single_stream'],3), 'glow_enc', round(p.get('conv_mfma.glow_encoder',0),3), 'elem', round(p.get('elementwise',0),3), 'half', round(h.get('utterances_per_sec',0),1), round(h.get('latency_ms_single_stream',0),3), 'c4', round(c['utterances_per_sec']), round(c['ms_per_call'],3), round(c.get('latency_ms_single_stream', c.get('ms_per_call_single_stream', 0)),3))"
}
for i in 1 2 3; do
  run base MI355TTS_LIN16_NO_LN=1
  run ln_fused X=1
done | tee $O/ab_lin16_ln.log
