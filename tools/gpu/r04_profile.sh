# r04 evidence recipe: tools/profile_round.sh r04 (kernel trace + PMC passes of the headline command and of config 4), the
# bench lines, the bf16x3 trace + PMC, the stress runs, the parity suite
bash tools/profile_round.sh r04 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
timeout 600 python bench.py --batch 8 --concurrency 2 --no-cpu-baseline --no-config3 --no-config4 --no-config5 > $O/bench_batch8.json 2> $O/bench_batch8.err
timeout 600 python bench.py --quality medium --no-cpu-baseline --no-config3 --no-config4 --no-config5 > $O/bench_medium.json 2> $O/bench_medium.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/bf16x3_trace -o trace --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --precision bf16x3 > $O/bf16x3_trace.log 2>&1
bash tools/pmc_half.sh r04/bf16x3_pmc > /dev/null 2>&1
rm -f $O/*/*_agent_info.csv $O/*/*/*_agent_info.csv
timeout 120 python tools/stress.py --unload-leg 10 2>&1 | tail -1 > $O/stress_unload.json
timeout 300 python tools/stress.py --calls 1500 --threads 6 2>&1 | tail -1 > $O/stress.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped" > $O/gputest_final.log
tail -n 2 $O/bench_n1.err; cat $O/stress_unload.json $O/gputest_final.log; cut -c1-300 $O/stress.json
