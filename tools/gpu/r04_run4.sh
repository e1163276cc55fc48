# r04 session 4: wave priority outside the main loop, fragment ring depth
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488 -DRB_NEW=1"
bash tools/gpu/rb_diag.sh r04_diag4 \
  "$S1 -DRB_PRIO=0" "$S1 -DRB_PRIO=3" "$S1 -DRB_PRIO=1" "$S1 -DRB_PRIO=3 -DMI355TTS_ARING=5" "$S1 -DRB_PRIO=0 -DMI355TTS_ARING=5" \
  "$S1 -DRB_ONLY=2 -DRB_PRIO=0" "$S1 -DRB_ONLY=2 -DRB_PRIO=3" "$S1 -DRB_ONLY=2 -DRB_PRIO=3 -DMI355TTS_ARING=5" "$S1 -DRB_ONLY=2 -DRB_PRIO=3 -DRB_ABL=31" > /dev/null
cat gpurun_out/r04_diag4/rb_diag.log
