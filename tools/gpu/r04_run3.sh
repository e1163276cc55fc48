# r04 session 3: ablations of the continuous-stream tile (where does a k = 3 tile's time go?)
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488 -DRB_NEW=1"
bash tools/gpu/rb_diag.sh r04_diag3 \
  "$S1 -DRB_ONLY=2" "$S1 -DRB_ONLY=2 -DRB_ABL=1" "$S1 -DRB_ONLY=2 -DRB_ABL=2" "$S1 -DRB_ONLY=2 -DRB_ABL=4" "$S1 -DRB_ONLY=2 -DRB_ABL=8" "$S1 -DRB_ONLY=2 -DRB_ABL=16" "$S1 -DRB_ONLY=2 -DRB_ABL=15" "$S1 -DRB_ONLY=2 -DRB_ABL=31" \
  "$S1" "$S1 -DRB_ABL=1" "$S1 -DRB_ABL=2" "$S1 -DRB_ABL=16" "$S1 -DRB_ABL=31" "$S1 -DRB_ONLY=0" "$S1 -DRB_ONLY=0 -DRB_ABL=31" > /dev/null
cat gpurun_out/r04_diag3/rb_diag.log
