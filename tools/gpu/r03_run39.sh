cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5"
run() {  # label, env, extra args
  env $2 timeout 400 $B $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d.get('half_mode',{})
print('$1', 'utt/s', round(d['value'],1), 'lat', round(d['latency_ms_single_stream'],3), 'half', round(h.get('utterances_per_sec',0),1))"
}
for i in 1 2; do
  run q4_c8 X=1 "--concurrency 8"
  run q4_c12 X=1 "--concurrency 12"
  run q4_c16 X=1 "--concurrency 16"
  run q8_c8 GPU_MAX_HW_QUEUES=8 "--concurrency 8"
  run q8_c16 GPU_MAX_HW_QUEUES=8 "--concurrency 16"
  run q2_c8 GPU_MAX_HW_QUEUES=2 "--concurrency 8"
done | tee $O/queue_sweep.log
