bash tools/gpu/mrf_variants.sh "-DMRF_K8 -DMRF_C=8 -DMRF_T=256 -DMRF_NW=2" "-DMRF_K8 -DMRF_C=8 -DMRF_T=512 -DMRF_NW=2"
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head
timeout 120 python tools/config4_probe.py 20 2>&1 | grep -E "config4|narrow|resblock"
