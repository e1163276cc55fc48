#!/bin/bash
# which hardware queue carries how many kernels at 8 calls in flight?  rocprofv3 --kernel-trace of a short headline run, BENCH_DUMMY_STREAMS = 0 / 3
set -u
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in ${@:-0 3}; do
  O=gpurun_out/r06_qmap/d$v; rm -rf $O; mkdir -p $O
  BENCH_DUMMY_STREAMS=$v timeout 300 rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --repeats 3 --no-config3 --no-config4 --no-config5 --no-cpu-baseline --no-micro-batch --no-half-mode --no-steady-state > $O/bench.log 2>&1
  python - $O/t_kernel_trace.csv $v <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("DUMMY", sys.argv[2], "columns", list(rows[0].keys()))
q = collections.Counter(r["Queue_Id"] for r in rows)
print(" kernels per queue:", dict(q))
big = collections.Counter(r["Queue_Id"] for r in rows if "rb_group" in r["Kernel_Name"] or "rb_pair" in r["Kernel_Name"])
print(" ResBlock launches per queue:", dict(big))
if "Stream_Id" in rows[0]:
    sq = collections.defaultdict(collections.Counter)
    for r in rows:
        sq[r["Stream_Id"]][r["Queue_Id"]] += 1
    print(" stream -> queue:", {s: dict(c) for s, c in sq.items()})
PY
  tail -1 $O/bench.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' value (traced)', round(d['value'],1))
except Exception as e: print(' bench line unreadable', e)
"
  rm -f $O/t_kernel_trace.csv $O/*agent_info.csv
done
