cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped" | tee $O/gputest13.log
timeout 120 python tools/stress.py --unload-leg 8 2>&1 | tail -3 | tee $O/stress_unload.json
timeout 600 python bench.py > $O/bench13.json 2> $O/bench13.err; tail -3 $O/bench13.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench13.json'))
print('value',d['value'],'ms',d['ms_per_step'],'lat',d['latency_ms_single_stream'],'frac',d['roofline']['frac'])
print('half',d['half_mode']['utterances_per_sec'], d['half_mode']['latency_ms_single_stream'])
print('c3',d['config3']['utterances_per_sec'],'c5',d['config5']['ms_to_first_audio'],d['config5']['x_realtime'])
c=d['config4']; print('c4',c['utterances_per_sec'],c['ms_per_call'],c['latency_ms_single_stream'],c['roofline']['narrow_stages']['frac'],c['roofline']['wide_stages']['frac'])
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['cores'],d['cpu_baseline']['rtf_1_thread'])
print(d['profile_ms_per_step']); print(d['per_rank'], d['host_affinity_rank0'])
PY
