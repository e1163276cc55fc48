cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
python -c "
import json
d=json.load(open('$O/bench_default.json')); h=d['half_mode']; c=d['config4']
print('utt/s', round(d['value'],1), 'lat', round(d['latency_ms_single_stream'],3), 'frac', round(d['roofline']['frac'],4), 'half', round(h['utterances_per_sec'],1), round(h['latency_ms_single_stream'],3), 'c3', round(d['config3']['utterances_per_sec'],1), 'c4', round(c['utterances_per_sec']), 'c5', round(d['config5']['ms_to_first_audio'],2), round(d['config5']['x_realtime']), 'cpu', round(d['cpu_baseline']['value'],2), 'coalesce', round(d['glow_coalescing']['utterances_per_sec'],1))"
