set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gputest3.log
timeout 120 python tools/config4_probe.py 10 > $O/c4_probe3.log 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $O/c4_trace3 -o trace --output-format csv -- python tools/config4_probe.py 10 > $O/c4_trace3.log 2>&1
rm -f $O/c4_trace3/*agent_info.csv
timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode > $O/bench_c4_3.json 2> $O/bench_c4_3.err
cat $O/gputest3.log $O/c4_probe3.log
