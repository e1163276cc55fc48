cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMRF_C=16 -DMRF_T=256 -DMRF_NW=4 tools/probe/mrf_bench.hip -o /tmp/mb 2>&1 | grep error
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMRF_C=16 -DMRF_T=256 -DMRF_NW=4 -DMRF_ABL=32 tools/probe/mrf_bench.hip -o /tmp/mb32 2>&1 | grep error
for f in 2 64 128 256 384 512 768 1024 1536 3072; do /tmp/mb $f; /tmp/mb32 $f; done 2>&1 | tee $O/mrf_sizes.log
