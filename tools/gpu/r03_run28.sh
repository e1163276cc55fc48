cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|skipped|assert" | head -8
