set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode > $O/bench_c4_new.json 2> $O/bench_c4_new.err
timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --set-option mrf_small=0 > $O/bench_c4_old.json 2> $O/bench_c4_old.err
timeout 120 python tools/config4_probe.py 10 > $O/c4_probe.log 2>&1
timeout 240 rocprofv3 --kernel-trace --stats -d $O/c4_trace -o trace --output-format csv -- python tools/config4_probe.py 10 > $O/c4_trace.log 2>&1
ls -R $O/c4_trace | head; rm -f $O/c4_trace/*/*_agent_info.csv
cat $O/c4_probe.log
