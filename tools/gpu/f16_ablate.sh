#!/bin/bash
# ablations of the fp16 conv tile (tools/probe/f16_bench.hip built with -DF16_ABLATE=<bits>): what a launch costs without its
# epilogue (1), MFMAs (2), weight loads (4), LDS reads (8), ring staging (16).  gpurun -- 'bash tools/gpu/f16_ablate.sh'
O=gpurun_out/f16_ablate
mkdir -p $O
for ab in 0 1 2 4 8 16 3 6 14 30 31; do
  B=tools/probe/f16_bench_ab$ab.bin
  [ -x $B ] || continue
  echo "=== ablate $ab"
  for K in 11 3; do $B 128 128 $K 3 39488 4 40 | grep -v "^check"; done
  $B 256 256 11 3 4936 4 40 | grep -v "^check"
  $B 128 128 0 3 39488 4 40 | grep -v "^check"
done 2>&1 | tee $O/ablate.txt
