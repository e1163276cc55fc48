cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_phase
timeout 600 python tools/phase_probe.py 8 40 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r04_phase/phase_probe.txt
