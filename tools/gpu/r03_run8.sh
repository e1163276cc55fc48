cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|SQC|TCP|TA|TD|GRBM|TCC)_[A-Z0-9_]+" | sort -u > $O/counters.txt
wc -l $O/counters.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMRF_C=16 -DMRF_T=256 -DMRF_NW=4 tools/probe/mrf_bench.hip -o /tmp/mb16
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_WAIT_IFETCH SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $O/mb_pmc_$n -o p --output-format csv -- /tmp/mb16 > $O/mb_pmc_$n.log 2>&1
  python tools/pmc_reduce.py $O/mb_pmc_$n/p_counter_collection.csv | grep mrf
done
