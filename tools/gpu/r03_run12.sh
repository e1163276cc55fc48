cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -s -k "bf16 or dropin or half" 2>&1 | grep -E "bf16x3|passed|failed|Error|error|LSB" | tee $O/gputest12.log
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tee -a $O/gputest12.log
