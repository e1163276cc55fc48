#!/bin/bash
# Collect the round's measurement evidence on the GPU box (run through gpurun):
#   kernel trace + stats of the bench command, HBM PMC passes (FETCH_SIZE and
#   WRITE_SIZE separately, as MI355X_MICROARCH.md prescribes), MFMA-busy PMC pass.
# Usage: tools/profile_round.sh r01
set -u
R=${1:-r05}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --no-micro-batch --no-steady-state --concurrency 1 --repeats 1"
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- $BENCH > $OUT/bench_trace.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- $BENCH > $OUT/bench_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write --output-format csv -- $BENCH > $OUT/bench_write.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc_sq -o sq --output-format csv -- $BENCH > $OUT/bench_sq.log 2>&1
# BASELINE config 4 (thorsten + 'medium', one padded batch of 8 per call): the same four passes over tools/config4_probe.py
C4="python tools/config4_probe.py 10"
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/medium_trace -o trace --output-format csv -- $C4 > $OUT/medium_trace.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/medium_pmc_fetch -o fetch --output-format csv -- $C4 > $OUT/medium_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/medium_pmc_write -o write --output-format csv -- $C4 > $OUT/medium_write.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/medium_pmc_sq -o sq --output-format csv -- $C4 > $OUT/medium_sq.log 2>&1
ls -R $OUT | head -60
# keep the merged-back payload small: counter CSVs can be large
for f in $OUT/*/*counter_collection.csv; do python tools/pmc_reduce.py $f > ${f%.csv}_by_kernel.csv; rm -f $f; done
rm -f $OUT/*/*_agent_info.csv
du -sh $OUT
