# attention_mfma_kernel on the standalone harness: exact (dk = 2 NK) and clamped variants, P = 33 .. 768
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
for P in 33 120 400 768; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATT_P=$P tools/probe/att_bench.hip -o /tmp/att_new_$P &
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATT_P=$P -DATT_EXACT=false tools/probe/att_bench.hip -o /tmp/att_gen_$P &
done
wait
for P in 33 120 400 768; do
  for w in new gen; do echo $w; /tmp/att_${w}_$P; done
done 2>&1 | tee $O/att_probe.log
