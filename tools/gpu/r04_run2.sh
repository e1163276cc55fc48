# r04 session 2: the continuous-stream 128-row tile (rb_conv.h) against the chunked one, per tap count and as a group
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488"
S0="-DCG_C=256 -DCG_L=4936"
bash tools/gpu/rb_diag.sh r04_diag2 \
  "$S1" "$S1 -DRB_NEW=1" "$S1 -DRB_NEW=1 -DMI355TTS_ARING=4" "$S1 -DRB_NEW=1 -DRB_LB=5" \
  "$S1 -DRB_ONLY=0" "$S1 -DRB_ONLY=1" "$S1 -DRB_ONLY=2" "$S1 -DRB_NEW=1 -DRB_ONLY=0" "$S1 -DRB_NEW=1 -DRB_ONLY=1" "$S1 -DRB_NEW=1 -DRB_ONLY=2" \
  "$S1 -DRB_NEW=1 -DCG_DIL=5" "$S0 -DRB_NEW=1" "$S0" > /dev/null
cat gpurun_out/r04_diag2/rb_diag.log
