# r04 session 1: diagnostics of the grouped f32 ResBlock launches (ramp / tail / streams / dispatch order / k-split tiles)
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488"
S0="-DCG_C=256 -DCG_L=4936"
T0="-DCG_CI=64 -DCG_MB=2 -DCG_NB=1 -DCG_WN=1 -DCG_KS=8 -DCG_WM=1"
bash tools/gpu/rb_diag.sh r04_diag1 \
  "$S1" "$S1 -DRB_CHUNK_STAMPS" "$S1 -DRB_ORDER=1" "$S1 -DCG_KS=2 -DCG_CI=32" "$S1 -DCG_KS=2 -DCG_CI=16" "$S1 -DMI355TTS_PROBE_NO_X2" "$S1 -DCG_CI=32" "$S1 -DCG_DIL=5" \
  "$S0 $T0" "$S0 $T0 -DRB_CHUNK_STAMPS" "$S0 $T0 -DRB_ORDER=1" "$S0 -DCG_KS=2 -DCG_CI=32" "$S0 -DCG_KS=2 -DCG_CI=16" "$S0 -DCG_CI=32 -DCG_NB=1 -DCG_KS=2" "$S0 $T0 -DMI355TTS_PROBE_NO_X2" > /dev/null
# the real pipeline under load: per-kernel begin / end of 8 calls in flight (do ResBlock launches of different streams overlap?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_diag1
timeout 300 rocprofv3 --kernel-trace -d $O/trace8 -o t --output-format csv -- python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --repeats 1 > $O/trace8.log 2>&1
python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/r04_diag1/trace8/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void mi355tts::", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
with open("gpurun_out/r04_diag1/trace8_compact.txt", "w") as o:
    t0 = rows[0][0]
    for s, e, n, q, st in rows:
        o.write(f"{s - t0} {e - t0} {q} {st} {n}\n")
print(len(rows), "kernels")
PY
rm -rf $O/trace8
tail -3 $O/trace8.log
cat $O/rb_diag.log
