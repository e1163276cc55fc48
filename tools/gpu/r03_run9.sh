bash tools/gpu/mrf_variants.sh "-DMRF_C=16 -DMRF_T=256 -DMRF_NW=4" "-DMRF_C=8 -DMRF_T=256 -DMRF_NW=4"
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|skipped"
timeout 120 python tools/config4_probe.py 20 2>&1 | grep -E "config4|narrow|resblock"
