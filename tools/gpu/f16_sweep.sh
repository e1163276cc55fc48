#!/bin/bash
# sweep of the native-f16 conv tile (tools/probe/f16_bench.hip) over the 'high' vocoder's layer shapes at the standard utterance
# (617 frames) and the tile configurations; gpurun -- 'bash tools/gpu/f16_sweep.sh'
O=gpurun_out/f16_sweep
mkdir -p $O
for v in "$@"; do
B=tools/probe/f16_bench_$v.bin
echo "=== variant $v"
for cfg in 0 1 4; do
  for K in 11 3; do $B 256 256 $K 3 4936 $cfg 30; done
  for K in 11 3; do $B 128 128 $K 3 39488 $cfg 30; done
  $B 256 256 0 3 4936 $cfg 30
  $B 128 128 0 3 39488 $cfg 30
done
for cfg in 2 5 0; do $B 64 64 0 3 78976 $cfg 30; done
for cfg in 3 6; do $B 32 32 0 3 157952 $cfg 30; done
for cfg in 0 4; do $B 512 256 2 1 617 $cfg 30 8; $B 256 128 2 1 4936 $cfg 30 8 1; done
$B 128 64 2 1 39488 0 30 2 1
$B 64 32 2 1 78976 2 30 2 1
done 2>&1 | grep -v "^check" | tee $O/sweep2.txt
