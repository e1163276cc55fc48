cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do
  for e in "X=1" "MI355TTS_UPS_NB2=256"; do
    env $e timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e', round(d['value'],1), round(d['latency_ms_single_stream'],3), round(d['profile_ms_per_step']['conv_mfma.hifigan_upsample'],3))"
  done
done | tee $O/ab_ups_nb2.log
