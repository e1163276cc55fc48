cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head
timeout 300 python -m pytest tests -m gpu -x -q -s -k "bf16" 2>&1 | grep -E "bf16x3|LSB"
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --precision bf16x3"
for i in 1 2; do
for e in "X=1" "MI355TTS_NO_BF16_UPS=1"; do
  env $e timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$e', round(d['value'],1), round(d['latency_ms_single_stream'],3), {k:round(v,3) for k,v in d['profile_ms_per_step'].items()})"
done; done | tee $O/ab_bf16_ups.log
