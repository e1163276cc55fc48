bash tools/profile_round.sh r03 2>&1 | tail -5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
timeout 120 python tools/stress.py --unload-leg 10 2>&1 | tail -1 > $O/stress_unload.json
tail -2 $O/bench_n1.err $O/bench_torchrun_n1.err; cat $O/stress_unload.json
