# compile + run variants of the MRF micro-benchmark on the GPU box: tools/gpu/mrf_variants.sh "<flags1>" "<flags2>" ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
i=0
for f in "$@"; do
  i=$((i+1))
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 $f -DMRF_TAG="\"$f\"" tools/probe/mrf_bench.hip -o /tmp/mrf_bench_$i 2>&1 | grep -E "error" ) &
done
wait
i=0
for f in "$@"; do
  i=$((i+1))
  /tmp/mrf_bench_$i
  /tmp/mrf_bench_$i 617
done 2>&1 | tee $O/mrf_variants.log
