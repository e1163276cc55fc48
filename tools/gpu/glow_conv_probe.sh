cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 $GC_FLAGS tools/probe/glow_conv_bench.hip -o /tmp/gcb && /tmp/gcb 2>&1 | tee $O/glow_conv_probe.log
