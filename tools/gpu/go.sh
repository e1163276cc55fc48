#!/bin/bash
# build the HIP library in-tree, then run a repo-relative script on an MI355X box: tools/gpu/go.sh <script> [gpurun timeout s]
set -e
cd "$(dirname "$0")/../.."
python -m larynx_amd.build 2>&1 | grep -E "error|Error" || true
/usr/local/graft/bin/gpurun --timeout ${2:-900} -- "bash $1" 2>&1 | tail -${3:-25}
