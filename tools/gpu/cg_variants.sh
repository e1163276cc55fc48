# compile + run variants of the conv_group micro-benchmark: tools/gpu/cg_variants.sh "<flags1>" "<flags2>" ...
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
i=0
for f in "$@"; do
  i=$((i+1))
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 $f tools/probe/conv_group_bench.hip -o /tmp/cgb_$i 2>&1 | grep -E "error" ) &
done
wait
i=0
for f in "$@"; do
  i=$((i+1))
  /tmp/cgb_$i; /tmp/cgb_$i
done 2>&1 | tee $O/cg_variants.log
