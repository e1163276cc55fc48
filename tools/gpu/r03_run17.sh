cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for e in "X=1" "MI355TTS_GLOW_TILES=512" "MI355TTS_GLOW_TILES=256" "MI355TTS_GLOW_TILES=128" "MI355TTS_GLOW_TILES=64"; do
  echo "== $e"
  env $e timeout 120 python tools/config4_probe.py 20 2>&1 | grep -E "config4|glow"
  env $e timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --steps 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config4']; print('  bench', round(d['value'],1), round(d['latency_ms_single_stream'],3), 'c4', round(c['utterances_per_sec'],0), round(c['latency_ms_single_stream'],3))"
done 2>&1 | tee $O/glow_tiles.log
