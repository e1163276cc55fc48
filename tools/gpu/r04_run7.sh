# r04 session 7: per-kernel durations of the two tiles in the pipeline (single stream, rocprofv3), then the parity suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab2; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1"
for v in old new old new; do
  opt=""; [ $v = old ] && opt="--set-option rb_conv=0"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t --output-format csv -- $B $opt > $O/t_$v.log 2>&1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -6 $f | cut -c1-160
  rm -rf $O/t_$v
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
