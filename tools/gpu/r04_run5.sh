# r04 session 5: SQ counters of the continuous-stream tile on the harness (what do the waves wait for?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pmc; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/counters.txt | sort -u > $O/sq_counters.txt
S1="-DCG_C=128 -DCG_L=39488 -DRB_NEW=1"
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 tools/probe/rb_diag.hip -o /tmp/rbd_g &
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 -DRB_ONLY=2 tools/probe/rb_diag.hip -o /tmp/rbd_k3 &
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 -DRB_ONLY=0 tools/probe/rb_diag.hip -o /tmp/rbd_k11 &
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 -DRB_ONLY=2 -DRB_ABL=31 tools/probe/rb_diag.hip -o /tmp/rbd_k3m &
wait
pass() { # name, counters...
  n=$1; shift
  for b in g k3 k11 k3m; do
    timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/${n}_$b -o p --output-format csv -- /tmp/rbd_$b /dev/null pmc > $O/${n}_$b.log 2>&1
    f=$(find $O/${n}_$b -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_reduce.py $f | grep -v "^kernel" | sed "s/^/$b,/" >> $O/${n}.csv; else tail -3 $O/${n}_$b.log; fi
    rm -rf $O/${n}_$b
  done
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass p2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass p3 SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass p4 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum
cat $O/p1.csv $O/p2.csv $O/p3.csv $O/p4.csv 2>/dev/null
wc -l $O/sq_counters.txt
