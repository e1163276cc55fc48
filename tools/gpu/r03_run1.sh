set -x
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gputest1.log
B="python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --no-config4 --quality medium"
timeout 300 $B --set-option mrf_small=0 > $O/medium_S_old.json 2> $O/medium_S_old.err
timeout 300 $B > $O/medium_S_new.json 2> $O/medium_S_new.err
timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode > $O/bench_c4_new.json 2> $O/bench_c4_new.err
timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --set-option mrf_small=0 > $O/bench_c4_old.json 2> $O/bench_c4_old.err
tail -3 $O/gputest1.log
