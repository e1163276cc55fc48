cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for c in 8 12 16; do
 for p in bf16x3 f32; do
  timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --precision $p --concurrency $c 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$p conc $c', round(d['value'],1), round(d['latency_ms_single_stream'],3))"
 done
done | tee $O/conc_sweep.log
