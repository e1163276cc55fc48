bash tools/profile_round.sh r03 2>&1 | tail -3
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
timeout 120 python tools/stress.py --unload-leg 10 2>&1 | tail -1 > $O/stress_unload.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/bf16x3_trace -o trace --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --precision bf16x3 > $O/bf16x3_trace.log 2>&1
rm -f $O/*/*_agent_info.csv
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped" > $O/gputest_final.log
timeout 300 python tools/stress.py --calls 1500 --threads 6 2>&1 | tail -1 > $O/stress.json
tail -n 2 $O/bench_n1.err; cat $O/stress_unload.json $O/gputest_final.log; cut -c1-300 $O/stress.json
