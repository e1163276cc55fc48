set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > $O/gputest4.log
for v in "NW4:" "NW8:MI355TTS_MRF_NW=8" "NW8T512:MI355TTS_MRF_NW=8 MI355TTS_MRF_T=512" "C8T256:MI355TTS_MRF_T=256"; do
  n=${v%%:*}; e=${v#*:}
  env $e timeout 120 python tools/config4_probe.py 20 2>&1 | grep -E "config4|narrow" > $O/c4_$n.log
done
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $O/c4_pmc_sq -o sq --output-format csv -- python tools/config4_probe.py 5 > $O/c4_pmc_sq.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/c4_pmc_sq2 -o sq --output-format csv -- python tools/config4_probe.py 5 > $O/c4_pmc_sq2.log 2>&1
for f in $O/c4_pmc_sq*/*counter_collection.csv; do python tools/pmc_reduce.py $f > ${f%.csv}_by_kernel.csv; rm -f $f; done
rm -f $O/*/*_agent_info.csv
cat $O/gputest4.log $O/c4_NW*.log $O/c4_C8*.log
