cd $GRAFT_REPO_ROOT
bash tools/gpu/rb_diag.sh r04_diag13 "-DCG_C=128 -DCG_L=39488 -DRB_NEW=1" > /dev/null
grep -E "small-launch|4 stream" gpurun_out/r04_diag13/rb_diag.log
