# r04 session 9: the harness under pipeline conditions (own planes per member, chained launches, rotating weights)
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488"
bash tools/gpu/rb_diag.sh r04_diag9 "$S1 -DRB_NEW=1" "$S1" > /dev/null
grep -E "^##|chained|1 stream|2 stream" gpurun_out/r04_diag9/rb_diag.log
