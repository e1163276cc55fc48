cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -5
B="python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode"
for i in 1 2; do
  for lib in libmi355tts_base.so libmi355tts.so; do
    timeout 300 $B --library larynx_amd/$lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['latency_ms_single_stream'],3), round(d['roofline']['frac'],4), round(d['profile_ms_per_step']['conv_mfma.hifigan_resblock'],3), 'c4', round(d['config4']['utterances_per_sec']), round(d['config4']['roofline']['wide_stages']['ms_per_call'],3))"
  done
done | tee $O/ab_pair_3wg.log
