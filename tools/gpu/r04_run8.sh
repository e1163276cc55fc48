# r04 session 8: per-position durations of the 18 grouped ResBlock launches of an utterance (single stream)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pos; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1"
timeout 300 rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- $B > $O/t.log 2>&1
python - <<'PY'
import csv, glob, re, collections
f = glob.glob("gpurun_out/r04_pos/t/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void mi355tts::", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
# split into utterances at conv_pre (conv_mfma_kernel<7, 64, 1, 1, 1, 8, 76, 0, 1>)... simpler: position index among group kernels since the last upsampler
pos = collections.defaultdict(list)
gap = collections.defaultdict(list)
i_in_stage = 0; stage = -1; prev_end = None
for s, e, n in rows:
    if "EPI" in n: pass
    if n.startswith("conv_mfma_kernel<2,") :  # an upsampler: new stage
        stage += 1; i_in_stage = 0
    elif n.startswith("conv_mfma_kernel<7, 64") and stage >= 3:
        stage = -1
    elif "group_kernel" in n:
        pos[(stage % 4, i_in_stage, n[:40])].append((e - s) / 1e3)
        if prev_end: gap[(stage % 4, i_in_stage)].append((s - prev_end) / 1e3)
        i_in_stage += 1
    prev_end = e
for k in sorted(pos):
    v = pos[k][len(pos[k]) // 4:]
    g = gap.get(k[:2], [0])
    print(k, "n=%d avg %.1f us min %.1f max %.1f  gap before %.1f us" % (len(v), sum(v) / len(v), min(v), max(v), sum(g) / len(g)))
PY
rm -rf $O/t
