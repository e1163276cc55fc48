cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -8
B="python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode"
run() {  # label, env
  env $2 timeout 400 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config4']
print('$1', 'utt/s', round(d['value'],1), 'lat', round(d['latency_ms_single_stream'],3), 'c4', round(c['utterances_per_sec']), round(c['ms_per_call'],3), round(c.get('latency_ms_single_stream', c.get('ms_per_call_single_stream', 0)),3))"
}
for i in 1 2 3; do
  run base MI355TTS_GATE16P_MIN=100000000
  run p384 X=1
  run p512 MI355TTS_GATE16P_WGS=512
  run p256 MI355TTS_GATE16P_WGS=256
done | tee $O/ab_gate16p.log
for m in 100000000 512; do MI355TTS_GATE16P_MIN=$m python tools/config4_probe.py 20 2>&1 | grep -E "config4:|glow_decoder"; done | tee -a $O/ab_gate16p.log
