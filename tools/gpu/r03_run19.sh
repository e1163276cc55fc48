cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do
  for lib in libmi355tts.so libmi355tts_prio1.so libmi355tts_prio3.so; do
    timeout 300 $B --library larynx_amd/$lib 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), round(d['latency_ms_single_stream'],3), round(d['roofline']['frac'],4), round(d['profile_ms_per_step']['conv_mfma.hifigan_resblock'],3))"
  done
done | tee $O/ab_setprio.log
