cd $GRAFT_REPO_ROOT
for c in 2 3 4 5 2 3 4 5; do timeout 200 python bench.py --no-cpu-baseline --steps 40 --concurrency $c >> gpurun_out/conc_$c.json 2>> gpurun_out/conc_$c.err; done
echo ok
