#!/bin/bash
# Round 5: per-kernel durations and kernels in flight under the headline's load (8 batch-1 calls in flight), for the two forms of
# the GlowTTS decoder's WaveNet layers (option wn_layer 0 / 1).  gpurun -- 'bash tools/gpu/r05_under_load.sh'
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r05_under_load
mkdir -p $O
for wn in 0 1; do
  timeout 300 rocprofv3 --kernel-trace -d $O/trace$wn -o t --output-format csv -- python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --no-steady-state --repeats 1 --set-option wn_layer=$wn > $O/trace$wn.log 2>&1
  python - $wn <<'PY'
import csv, glob, re, sys
wn = sys.argv[1]
f = glob.glob(f"gpurun_out/r05_under_load/trace{wn}/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void mi355tts::", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
with open(f"gpurun_out/r05_under_load/compact{wn}.txt", "w") as o:
    t0 = rows[0][0]
    for s, e, n, q, st in rows:
        o.write(f"{s - t0} {e - t0} {q} {st} {n}\n")
print(len(rows), "kernels")
PY
  rm -rf $O/trace$wn
  python tools/overlap_report.py $O/compact$wn.txt gate16,wn_layer > $O/report$wn.txt 2>&1
  rm -f $O/compact$wn.txt
done
tail -3 $O/trace0.log | cut -c1-300
