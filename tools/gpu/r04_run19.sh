# r04 session 19: the upsamplers on the continuous-stream tile: per-kernel durations (single stream) and the bench line, A B A B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab9; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1"
for v in old new; do
  opt=""; [ $v = old ] && opt="--set-option rb_conv=0"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t --output-format csv -- $B $opt > $O/t_$v.log 2>&1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "conv_mfma_kernel<2,|rb_conv_kernel|conv_mfma_kernel<7, 32" $f | cut -c1-150
  rm -rf $O/t_$v
done
B2="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do
  timeout 300 $B2 --set-option rb_conv=0 > $O/old_$i.json 2> $O/old_$i.err
  timeout 300 $B2 > $O/new_$i.json 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab9/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); p = j["profile_ms_per_step"]
    print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"], "resblock %.3f ups %.3f prepost %.3f" % (p["conv_mfma.hifigan_resblock"], p["conv_mfma.hifigan_upsample"], p["conv_mfma.hifigan_pre_post"]))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed"
