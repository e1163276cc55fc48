set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" > $O/gputest5.log
for v in "A:" "B:MI355TTS_MRF_T=256"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 120 python tools/config4_probe.py 20 2>&1 | grep -E "config4|narrow|resblock" > $O/c4_5$n.log
done
timeout 240 rocprofv3 --kernel-trace --stats -d $O/c4_trace5 -o trace --output-format csv -- python tools/config4_probe.py 10 > $O/c4_trace5.log 2>&1
rm -f $O/c4_trace5/*agent_info.csv
timeout 300 python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --no-config4 --quality medium > $O/medium_S_5.json 2> $O/medium_S_5.err
cat $O/gputest5.log $O/c4_5*.log
grep mrf_small $O/c4_trace5/trace_kernel_stats.csv
