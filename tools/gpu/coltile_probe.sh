cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 $CT_FLAGS tools/probe/coltile_bench.hip -o /tmp/ctb 2>/dev/null && /tmp/ctb 2>&1 | tee $O/coltile_probe.log
