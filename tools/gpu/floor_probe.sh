cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/probe/launch_floor.hip -o /tmp/launch_floor && { /tmp/launch_floor; /tmp/launch_floor; } 2>&1 | tee $O/launch_floor.log
