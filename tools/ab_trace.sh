#!/bin/bash
# A/B two builds of the library by per-kernel rocprofv3 averages, inside ONE gpurun call and in
# A B A B order (boxes differ by a few percent and drift by ~2 % within a session, so neither
# wall-clock A/B across calls nor a single A-then-B pass resolves 1-3 % effects):
#   gpurun -- 'bash tools/ab_trace.sh larynx_amd/lib_a.so larynx_amd/lib_b.so' ; python tools/ab_trace_report.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ab[0-9]*
i=0
for l in "$1" "$2" "$1" "$2"; do
  MI355TTS_LIB=$l timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ab$i -o t --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial-branches --concurrency 1 > gpurun_out/ab$i.log 2>&1
  find gpurun_out/ab$i -name "*kernel_trace.csv" -delete
  i=$((i+1))
done
echo ok
