#!/bin/bash
# SQ counters of the native fp16 mode, per kernel (run through gpurun).  Usage: [ENV=..] tools/pmc_half.sh [tag]
set -u
R=${1:-f16pmc}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-config3 --no-config5 --no-half-mode --no-micro-batch --no-steady-state --concurrency 1 --repeats 1 --precision f16"
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc_sq -o sq --output-format csv -- $BENCH > $OUT/bench_sq.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA -d $OUT/pmc_sq2 -o sq2 --output-format csv -- $BENCH > $OUT/bench_sq2.log 2>&1
for f in $OUT/*/*counter_collection.csv; do python tools/pmc_reduce.py $f --split-workgroups "conv_f16_group_kernel<11, 7, 3, 2, 2, 2, 2" 500 > ${f%.csv}_by_kernel.csv; rm -f $f; done
rm -f $OUT/*/*_agent_info.csv
grep -h "f16" $OUT/*/*_by_kernel.csv | cut -c1-200
