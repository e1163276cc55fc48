cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
B="-DMRF_C=16 -DMRF_T=256 -DMRF_NW=4"
hipcc --offload-arch=gfx950 -O3 -std=c++17 $B -DMRF_ABL=72 tools/probe/mrf_bench.hip -o /tmp/mb72 2>&1 | grep error
hipcc --offload-arch=gfx950 -O3 -std=c++17 $B -DMRF_ABL=78 tools/probe/mrf_bench.hip -o /tmp/mb78 2>&1 | grep error
for f in 2 128 256 384 768 1536; do /tmp/mb72 $f; /tmp/mb78 $f; done 2>&1 | tee $O/mrf_sizes2.log
/tmp/mb72; /tmp/mb78
hipcc --offload-arch=gfx950 -O3 -std=c++17 $B -DMRF_ABL=72 -S --cuda-device-only tools/probe/mrf_bench.hip -o $O/mb72.s 2>&1 | grep error
