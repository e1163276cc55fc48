#!/bin/bash
# Round-6 final evidence with the final build (ONE gpurun call): rocprofv3 passes of the f32 headline and config 4 (tools/profile_round.sh r06),
# of the fp16 mode (kernel trace + two SQ passes), the bench lines (default, the driver's exact flags, the driver's torchrun form, --precision f16),
# the -m gpu suite and smoke().  Reduce afterwards: tools/profile_summary.py r06; ... r06 medium; tools/f16_summary.py r06_f16.
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "rc $?" >> $O/gputest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 \
  > $O/bench_torchrun_n1.json 2> $O/bench_torchrun.err
timeout 600 python bench.py --precision f16 --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch > $O/bench_f16.json 2> $O/bench_f16.err
tools/profile_round.sh r06 > $O/profile_round.log 2>&1
F=gpurun_out/r06_f16
rm -rf $F; mkdir -p $F
F16="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --no-micro-batch --no-steady-state --concurrency 1 --repeats 1 --precision f16"
timeout 240 rocprofv3 --kernel-trace --stats -d $F/trace -o trace --output-format csv -- $F16 > $F/bench_trace.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $F/pmc_sq -o sq --output-format csv -- $F16 > $F/bench_sq.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA -d $F/pmc_sq2 -o sq2 --output-format csv -- $F16 > $F/bench_sq2.log 2>&1
for f in $F/*/*counter_collection.csv; do python tools/pmc_reduce.py $f --split-workgroups "conv_f16_group_kernel<11, 7, 3, 2, 2, 2, 2" 500 > ${f%.csv}_by_kernel.csv; rm -f $f; done
rm -f $F/*/*_agent_info.csv
tail -3 $O/gputest.log; cat $O/smoke.log | tail -5
for f in bench_n1 bench_driver_cmd bench_torchrun_n1 bench_f16; do python - $O/$f.json $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    h = d.get("half_mode") or {}
    print(sys.argv[2], "value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3), "half", round(h.get("utterances_per_sec", 0), 1),
          "steady", round((d.get("steady_state") or {}).get("utterances_per_sec", 0), 1), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "unreadable:", e)
PY
done
du -sh gpurun_out/r06 gpurun_out/r06_f16 $O
