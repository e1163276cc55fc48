set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" > $O/gputest6.log
for v in "A:" "B:MI355TTS_MRF_T=256"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 120 python tools/config4_probe.py 20 2>&1 | grep -E "config4|narrow|resblock" > $O/c4_6$n.log
done
timeout 240 rocprofv3 --kernel-trace --stats -d $O/c4_trace6 -o trace --output-format csv -- python tools/config4_probe.py 10 > $O/c4_trace6.log 2>&1
rm -f $O/c4_trace6/*agent_info.csv
cat $O/gputest6.log $O/c4_6*.log
grep mrf_small $O/c4_trace6/trace_kernel_stats.csv
