#!/bin/bash
# The gpurun sessions of round 4, one function per session (the record of what produced profiles/r04_*):
#   /usr/local/graft/bin/gpurun -- "bash tools/gpu/r04_sessions.sh <session>"
# Sessions that A/B library builds name .so files that existed only during the session (the -D switch is in the script).
set -u

diag_streams() {
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
}

diag_new_tile() {
# r04 session 2: the continuous-stream 128-row tile (rb_conv.h) against the chunked one, per tap count and as a group
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488"
S0="-DCG_C=256 -DCG_L=4936"
bash tools/gpu/rb_diag.sh r04_diag2 \
  "$S1" "$S1 -DRB_NEW=1" "$S1 -DRB_NEW=1 -DMI355TTS_ARING=4" "$S1 -DRB_NEW=1 -DRB_LB=5" \
  "$S1 -DRB_ONLY=0" "$S1 -DRB_ONLY=1" "$S1 -DRB_ONLY=2" "$S1 -DRB_NEW=1 -DRB_ONLY=0" "$S1 -DRB_NEW=1 -DRB_ONLY=1" "$S1 -DRB_NEW=1 -DRB_ONLY=2" \
  "$S1 -DRB_NEW=1 -DCG_DIL=5" "$S0 -DRB_NEW=1" "$S0" > /dev/null
cat gpurun_out/r04_diag2/rb_diag.log
}

diag_ablations() {
# r04 session 3: ablations of the continuous-stream tile (where does a k = 3 tile's time go?)
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488 -DRB_NEW=1"
bash tools/gpu/rb_diag.sh r04_diag3 \
  "$S1 -DRB_ONLY=2" "$S1 -DRB_ONLY=2 -DRB_ABL=1" "$S1 -DRB_ONLY=2 -DRB_ABL=2" "$S1 -DRB_ONLY=2 -DRB_ABL=4" "$S1 -DRB_ONLY=2 -DRB_ABL=8" "$S1 -DRB_ONLY=2 -DRB_ABL=16" "$S1 -DRB_ONLY=2 -DRB_ABL=15" "$S1 -DRB_ONLY=2 -DRB_ABL=31" \
  "$S1" "$S1 -DRB_ABL=1" "$S1 -DRB_ABL=2" "$S1 -DRB_ABL=16" "$S1 -DRB_ABL=31" "$S1 -DRB_ONLY=0" "$S1 -DRB_ONLY=0 -DRB_ABL=31" > /dev/null
cat gpurun_out/r04_diag3/rb_diag.log
}

diag_setprio() {
# r04 session 4: wave priority outside the main loop, fragment ring depth
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488 -DRB_NEW=1"
bash tools/gpu/rb_diag.sh r04_diag4 \
  "$S1 -DRB_PRIO=0" "$S1 -DRB_PRIO=3" "$S1 -DRB_PRIO=1" "$S1 -DRB_PRIO=3 -DMI355TTS_ARING=5" "$S1 -DRB_PRIO=0 -DMI355TTS_ARING=5" \
  "$S1 -DRB_ONLY=2 -DRB_PRIO=0" "$S1 -DRB_ONLY=2 -DRB_PRIO=3" "$S1 -DRB_ONLY=2 -DRB_PRIO=3 -DMI355TTS_ARING=5" "$S1 -DRB_ONLY=2 -DRB_PRIO=3 -DRB_ABL=31" > /dev/null
cat gpurun_out/r04_diag4/rb_diag.log
}

harness_pmc() {
# r04 session 5: SQ counters of the continuous-stream tile on the harness (what do the waves wait for?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pmc; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $O/counters.txt | sort -u > $O/sq_counters.txt
S1="-DCG_C=128 -DCG_L=39488 -DRB_NEW=1"
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 tools/probe/rb_diag.hip -o /tmp/rbd_g &
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 -DRB_ONLY=2 tools/probe/rb_diag.hip -o /tmp/rbd_k3 &
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 -DRB_ONLY=0 tools/probe/rb_diag.hip -o /tmp/rbd_k11 &
hipcc --offload-arch=gfx950 -O3 -std=c++17 $S1 -DRB_ONLY=2 -DRB_ABL=31 tools/probe/rb_diag.hip -o /tmp/rbd_k3m &
wait
pass() { # name, counters...
  n=$1; shift
  for b in g k3 k11 k3m; do
    timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/${n}_$b -o p --output-format csv -- /tmp/rbd_$b /dev/null pmc > $O/${n}_$b.log 2>&1
    f=$(find $O/${n}_$b -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python tools/pmc_reduce.py $f | grep -v "^kernel" | sed "s/^/$b,/" >> $O/${n}.csv; else tail -3 $O/${n}_$b.log; fi
    rm -rf $O/${n}_$b
  done
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS
pass p2 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass p3 SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass p4 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum
cat $O/p1.csv $O/p2.csv $O/p3.csv $O/p4.csv 2>/dev/null
wc -l $O/sq_counters.txt
}

ab_rb_pipeline() {
# r04 session 6: the continuous-stream tile in the pipeline (A B A B inside one session) + the parity suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab1; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do
  timeout 300 $B --set-option rb_conv=0 > $O/old_$i.json 2> $O/old_$i.err
  timeout 300 $B > $O/new_$i.json 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab1/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j["profile_ms_per_step"]
        print(f.split("/")[-1], "value %.1f" % j["value"], "latency %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"],
              "resblock %.3f" % p["conv_mfma.hifigan_resblock"], "ups %.3f" % p["conv_mfma.hifigan_upsample"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5
}

ab_rb_kernel_stats() {
# r04 session 7: per-kernel durations of the two tiles in the pipeline (single stream, rocprofv3), then the parity suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab2; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1"
for v in old new old new; do
  opt=""; [ $v = old ] && opt="--set-option rb_conv=0"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t --output-format csv -- $B $opt > $O/t_$v.log 2>&1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; head -6 $f | cut -c1-160
  rm -rf $O/t_$v
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
}

per_position() {
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
}

diag_pipeline_conditions() {
# r04 session 9: the harness under pipeline conditions (own planes per member, chained launches, rotating weights)
cd $GRAFT_REPO_ROOT
S1="-DCG_C=128 -DCG_L=39488"
bash tools/gpu/rb_diag.sh r04_diag9 "$S1 -DRB_NEW=1" "$S1" > /dev/null
grep -E "^##|chained|1 stream|2 stream" gpurun_out/r04_diag9/rb_diag.log
}

ab_stage0_m128() {
# r04 session 10: bench line of the current build (new fields), 128-row tile for stage 0 too, calls in flight x GlowTTS coalescing
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab3; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j["profile_ms_per_step"]; gc = j.get("glow_coalescing") or {}
        print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"],
              "resblock %.3f" % p["conv_mfma.hifigan_resblock"], "glow_under_load %s" % j.get("glow_under_load_ms"), "voc_only %s" % (j.get("vocoder_only_under_load") or {}).get("utterances_per_sec"),
              "host_cpu_ms %s" % j.get("host_cpu_ms_per_utterance"), "coalesced %.1f (%.1f rows)" % (gc.get("utterances_per_sec", 0), gc.get("rows_per_pass", 0)))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
}
for i in 1 2; do
  timeout 300 $B > $O/base_$i.json 2> $O/base_$i.err
  MI355TTS_M128_MIN_TILES=64 timeout 300 $B > $O/m128s0_$i.json 2> $O/m128s0_$i.err
done

show $O/base_1.json $O/m128s0_1.json $O/base_2.json $O/m128s0_2.json
}

ab_glow_priority_stream() {
# r04 session 11: GlowTTS on a high-priority stream (option glow_priority), hardware queue counts
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab4; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j["profile_ms_per_step"]; gc = j.get("glow_coalescing") or {}
        print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "glow_under_load %.3f" % (j.get("glow_under_load_ms") or 0),
              "voc_only %.1f" % ((j.get("vocoder_only_under_load") or {}).get("utterances_per_sec") or 0), "coalesced %.1f (%.1f rows)" % (gc.get("utterances_per_sec", 0), gc.get("rows_per_pass", 0)))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
}
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err; }
run base_1 $B
run prio_1 $B --set-option glow_priority=1
run base_2 $B
run prio_2 $B --set-option glow_priority=1
run low_1 $B --set-option glow_priority=2
GPU_MAX_HW_QUEUES=8 run q8 $B
GPU_MAX_HW_QUEUES=8 run q8_prio $B --set-option glow_priority=1
GPU_MAX_HW_QUEUES=6 run q6_prio $B --set-option glow_priority=1
run prio_c12 $B --set-option glow_priority=1 --concurrency 12
MI355TTS_M128_MIN_TILES=64 run prio_m128 $B --set-option glow_priority=1
show $O/base_1.json $O/prio_1.json $O/base_2.json $O/prio_2.json $O/low_1.json $O/q8.json $O/q8_prio.json $O/q6_prio.json $O/prio_c12.json $O/prio_m128.json
}

phase_probe() {
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_phase
timeout 600 python tools/phase_probe.py 8 40 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r04_phase/phase_probe.txt
}

small_launches() {
cd $GRAFT_REPO_ROOT
bash tools/gpu/rb_diag.sh r04_diag13 "-DCG_C=128 -DCG_L=39488 -DRB_NEW=1" > /dev/null
grep -E "small-launch|4 stream" gpurun_out/r04_diag13/rb_diag.log
}

stage0_ablation() {
cd $GRAFT_REPO_ROOT
S0="-DCG_C=256 -DCG_L=4936 -DCG_CI=64 -DCG_MB=2 -DCG_NB=1 -DCG_WN=1 -DCG_KS=8 -DCG_WM=1"
bash tools/gpu/rb_diag.sh r04_diag14 "$S0" "$S0 -DMI355TTS_ABLATION -DCG_ABLATE=5" "$S0 -DMI355TTS_ABLATION -DCG_ABLATE=7" "$S0 -DMI355TTS_ABLATION -DCG_ABLATE=2" "$S0 -DRB_ONLY=2" "$S0 -DRB_ONLY=2 -DMI355TTS_ABLATION -DCG_ABLATE=5" "$S0 -DRB_ONLY=0" "$S0 -DRB_ONLY=0 -DMI355TTS_ABLATION -DCG_ABLATE=5"> /dev/null
grep -E "^##|L x8|1 stream|2 stream" gpurun_out/r04_diag14/rb_diag.log
}

ab_glow_lds_footprint() {
# r04 session 15: LDS footprint of the GlowTTS launches under load — 16-row conv tiles at 30 KB (halo 8) vs 36 KB (halo 16),
# attention at 66 KB vs 144 KB
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab5; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j["profile_ms_per_step"]
        print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "glow_under_load %.3f" % (j.get("glow_under_load_ms") or 0),
              "voc_only %.1f" % ((j.get("vocoder_only_under_load") or {}).get("utterances_per_sec") or 0), "dec %.3f enc %.3f elem %.3f" % (p["conv_mfma.glow_decoder"], p["conv_mfma.glow_encoder"], p["elementwise"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
}
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err; }
for i in 1 2; do
  MI355TTS_ATT_BIG_LDS=1 run h16_$i $B --library larynx_amd/libmi355tts_h16.so
  MI355TTS_ATT_BIG_LDS=1 run h8_attbig_$i $B
  run h8_$i $B
done
MI355TTS_M128_MIN_TILES=64 run h8_m128 $B
show $O/h16_1.json $O/h8_attbig_1.json $O/h8_1.json $O/h16_2.json $O/h8_attbig_2.json $O/h8_2.json $O/h8_m128.json
timeout 300 python tools/phase_probe.py 8 40 2>&1 | grep "threads, GlowTTS then" 
}

ab_glow_wave_priority() {
# r04 session 16: s_setprio 3 at the entry of the GlowTTS kernels vs none
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab6; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        p = j["profile_ms_per_step"]
        print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "glow_under_load %.3f" % (j.get("glow_under_load_ms") or 0),
              "voc_only %.1f" % ((j.get("vocoder_only_under_load") or {}).get("utterances_per_sec") or 0), "dec %.3f enc %.3f elem %.3f" % (p["conv_mfma.glow_decoder"], p["conv_mfma.glow_encoder"], p["elementwise"]))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
}
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err; }
for i in 1 2; do
  run noprio_$i $B --library larynx_amd/libmi355tts_noprio.so
  run prio_$i $B
done
MI355TTS_M128_MIN_TILES=64 run prio_m128 $B
show $O/noprio_1.json $O/prio_1.json $O/noprio_2.json $O/prio_2.json $O/prio_m128.json
timeout 300 python tools/phase_probe.py 8 40 2>&1 | grep "threads, GlowTTS then" 
}

ab_host_wait() {
# r04 session 17: how the caller threads wait for their streams (spin / blocking event / query + sleep)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab7; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "glow_under_load %.3f" % (j.get("glow_under_load_ms") or 0),
              "voc_only %.1f" % ((j.get("vocoder_only_under_load") or {}).get("utterances_per_sec") or 0), "host_cpu_ms %.1f" % j["host_cpu_ms_per_utterance"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
}
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err; }
for i in 1 2; do
  MI355TTS_SYNC_MODE=0 run spin_$i $B
  MI355TTS_SYNC_MODE=1 run block_$i $B
  MI355TTS_SYNC_MODE=2 run poll_$i $B
done
MI355TTS_SYNC_MODE=1 run block_c16 $B --concurrency 16
MI355TTS_SYNC_MODE=2 run poll_c16 $B --concurrency 16
show $O/spin_1.json $O/block_1.json $O/poll_1.json $O/spin_2.json $O/block_2.json $O/poll_2.json $O/block_c16.json $O/poll_c16.json
}

ab_calls_in_flight() {
# r04 session 18: more calls in flight with coalesced GlowTTS passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab8; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
run() { n=$1; shift; timeout 400 "$@" > $O/$n.json 2> $O/$n.err; }
for c in 8 16 24 32; do run c$c $B --concurrency $c; done
MI355TTS_SYNC_MODE=2 run c32_poll $B --concurrency 32
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab8/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); gc = j.get("glow_coalescing") or {}
        print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "glow_under_load %.3f" % (j.get("glow_under_load_ms") or 0),
              "voc_only %.1f" % ((j.get("vocoder_only_under_load") or {}).get("utterances_per_sec") or 0), "coalesced %.1f (%.1f rows)" % (gc.get("utterances_per_sec", 0), gc.get("rows_per_pass", 0)), "host_cpu %.1f" % j["host_cpu_ms_per_utterance"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-300:])
PY
}

ab_rb_upsamplers() {
# r04 session 19: the upsamplers on the continuous-stream tile: per-kernel durations (single stream) and the bench line, A B A B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab9; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1"
for v in old new; do
  opt=""; [ $v = old ] && opt="--set-option rb_conv=0"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t --output-format csv -- $B $opt > $O/t_$v.log 2>&1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "conv_mfma_kernel<2,|rb_conv_kernel|conv_mfma_kernel<7, 32" $f | cut -c1-150
  rm -rf $O/t_$v
done
B2="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do
  timeout 300 $B2 --set-option rb_conv=0 > $O/old_$i.json 2> $O/old_$i.err
  timeout 300 $B2 > $O/new_$i.json 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab9/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); p = j["profile_ms_per_step"]
    print(f.split("/")[-1], "value %.1f" % j["value"], "lat %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"], "resblock %.3f ups %.3f prepost %.3f" % (p["conv_mfma.hifigan_resblock"], p["conv_mfma.hifigan_upsample"], p["conv_mfma.hifigan_pre_post"]))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed"
}

ab_voc_only_sleep() {
# the vocoder-only region of bench.py with a HOST sleep in front of every call (an acoustic pass that takes its under-load
# latency but no GPU): how much of glow_under_load_ms is latency (callers not feeding the GPU), how much is GPU work?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab10; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for c in 8 16; do for ms in 0 4 8 12; do timeout 300 $B --concurrency $c --voc-only-sleep-ms $ms > $O/c${c}_s$ms.json 2> $O/c${c}_s$ms.err; done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab10/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1][:-5], "full call %.1f /s" % j["value"], "vocoder-only region (with the sleep) %.1f /s" % j["vocoder_only_under_load"]["utterances_per_sec"])
PY
}

ab_steps() {
# value against the number of steps K of a timed region (each region starts on an idle GPU and drains at its end)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab11; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for k in 20 50 100 200 400; do timeout 600 $B --steps $k > $O/k$k.json 2> $O/k$k.err; done
timeout 600 $B --steps 200 --concurrency 12 > $O/k200_c12.json 2> $O/k200_c12.err
timeout 600 $B --steps 200 --concurrency 16 > $O/k200_c16.json 2> $O/k200_c16.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab11/*.json"), key=lambda x: (len(x), x)):
    j = json.loads(open(f).read().strip().splitlines()[-1]); gc = j.get("glow_coalescing") or {}
    print(f.split("/")[-1][:-5], "value %.1f /s" % j["value"], "ms_per_step %.3f" % j["ms_per_step"], "vocoder-only %.1f /s" % j["vocoder_only_under_load"]["utterances_per_sec"], "glow_under_load %.3f" % j["glow_under_load_ms"],
          "latency %.3f" % j["latency_ms_single_stream"], "repeats %d" % j["timing"]["repeats"], "coalesced %.1f (%.1f rows)" % (gc.get("utterances_per_sec", 0), gc.get("rows_per_pass", 0)))
PY
}

ab_stage0_m128_steady() {
# the 128-row continuous-stream tile for the 256-channel stage too (MI355TTS_M128_MIN_TILES=64), read in the steady-state leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab12; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do
  timeout 600 $B > $O/base_$i.json 2> $O/base_$i.err
  MI355TTS_M128_MIN_TILES=64 timeout 600 $B > $O/m128_$i.json 2> $O/m128_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab12/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1][:-5], "value %.1f" % j["value"], "steady %.1f" % j["steady_state"]["utterances_per_sec"], "vocoder-only %.1f" % j["vocoder_only_under_load"]["utterances_per_sec"], "latency %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"])
PY
}

ab_rb_pair() {
# the 4-wave fused pair kernel without a k-split (rb_pair.h, option rb_pair) vs the 8-wave one: kernel stats, bench, parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab13; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --no-steady-state"
for v in old new; do
  opt=""; [ $v = old ] && opt="--set-option rb_pair=0"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t --output-format csv -- $B $opt > $O/t_$v.log 2>&1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "pair" $f | cut -c1-150
  rm -rf $O/t_$v
done
B2="python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode"
for i in 1 2; do
  timeout 300 $B2 --set-option rb_pair=0 > $O/old_$i.json 2> $O/old_$i.err
  timeout 300 $B2 > $O/new_$i.json 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab13/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); p = j["profile_ms_per_step"]
    print(f.split("/")[-1][:-5], "value %.1f" % j["value"], "steady %.1f" % j["steady_state"]["utterances_per_sec"], "lat %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"],
          "resblock %.3f" % p["conv_mfma.hifigan_resblock"], "config4 %.0f (%.3f ms single)" % (j["config4"]["utterances_per_sec"], j["config4"]["latency_ms_single_stream"]))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
}

ab_rb_pair_variants() {
# rb_pair.h variants: conv2's bias / residual requested before its MFMA phase (-DRBP_PREFETCH64 / 32), the C = 32 kernel at
# three waves per SIMD (-DRBP_LB32=3); libraries built with those switches next to the default one
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab14; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --no-steady-state"
for v in "" _p64 _lb3 _p64_p32lb3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t$v -o t --output-format csv -- $B --library larynx_amd/libmi355tts$v.so > $O/t$v.log 2>&1
  f=$(find $O/t$v -name "*kernel_stats.csv" | head -1)
  echo "== lib$v"; grep -E "pair" $f | cut -c1-140
  rm -rf $O/t$v
done
B2="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do for v in "" _p64 _lb3 _p64_p32lb3; do timeout 300 $B2 --library larynx_amd/libmi355tts$v.so > $O/lib${v}_$i.json 2> $O/lib${v}_$i.err; done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab14/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); p = j["profile_ms_per_step"]
    print(f.split("/")[-1][:-5], "value %.1f" % j["value"], "steady %.1f" % j["steady_state"]["utterances_per_sec"], "lat %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"], "resblock %.3f" % p["conv_mfma.hifigan_resblock"])
PY
}

diag_stage0_nb1() {
# the 256-channel stage on the continuous-stream tile with ONE column block per wave (128 rows x 32 columns: 930 workgroups)
cd $GRAFT_REPO_ROOT
S0="-DCG_C=256 -DCG_L=4936"
T0="-DCG_CI=64 -DCG_MB=2 -DCG_NB=1 -DCG_WN=1 -DCG_KS=8 -DCG_WM=1"
bash tools/gpu/rb_diag.sh r04_diag15 "$S0 $T0" "$S0 -DRB_NEW=1 -DCG_NB=1" "$S0 -DRB_NEW=1 -DCG_NB=1 -DRB_LB=5" "$S0 -DRB_NEW=1 -DCG_NB=2" "$S0 -DRB_NEW=1 -DCG_NB=1 -DRB_ONLY=2" "$S0 $T0 -DRB_ONLY=2" "-DCG_C=128 -DCG_L=39488 -DRB_NEW=1 -DCG_NB=1" > /dev/null
grep -E "^##|member k|L x8|1 stream|2 stream|4 stream" gpurun_out/r04_diag15/rb_diag.log
}


ab_wide_stores() {
# interior tiles of rb_conv.h / rb_pair.h store 16 bytes per lane after a transpose through LDS (-DRB_WIDE_STORES / -DRBP_WIDE_STORES,
# default 1).  Three builds: _nowide (both 0), _rbwide (rb_conv.h only), default (both).  Bits, kernel stats, bench, parity.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab15; mkdir -p $O
for q in high medium; do PYTHONPATH=. python tools/ab_bits.py larynx_amd/libmi355tts_nowide.so larynx_amd/libmi355tts.so $q; done
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --no-steady-state"
for v in _nowide _rbwide ""; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/t$v -o t --output-format csv -- $B --library larynx_amd/libmi355tts$v.so > $O/t$v.log 2>&1
  f=$(find $O/t$v -name "*kernel_stats.csv" | head -1)
  echo "== lib$v"; grep -E "rb_|conv_group|conv_kernel" $f | cut -c1-150
  rm -rf $O/t$v
done
B2="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode"
for i in 1 2; do for v in _nowide _rbwide ""; do timeout 300 $B2 --library larynx_amd/libmi355tts$v.so > $O/lib${v}_$i.json 2> $O/lib${v}_$i.err; done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab15/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); p = j["profile_ms_per_step"]
    print(f.split("/")[-1][:-5], "value %.1f" % j["value"], "steady %.1f" % j["steady_state"]["utterances_per_sec"], "lat %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"], "resblock %.3f" % p["conv_mfma.hifigan_resblock"])
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
}

ab_wide_stores_inproc() {
# the same three builds alternating inside ONE process (tools/ab_inproc.py): single-stream and four caller threads
cd $GRAFT_REPO_ROOT
L="larynx_amd/libmi355tts_nowide.so larynx_amd/libmi355tts_rbwide.so larynx_amd/libmi355tts.so"
PYTHONPATH=. timeout 400 python tools/ab_inproc.py --streams 1 $L
PYTHONPATH=. timeout 400 python tools/ab_inproc.py --streams 4 --calls 10 $L
PYTHONPATH=. timeout 400 python tools/ab_inproc.py --streams 1 $L
}

diag_snake_order() {
# the 256-channel stage on the continuous-stream tile (128 x 64, 468 workgroups at 624 frames: all resident at once) with the
# rounds of the dispatch laid out as a snake (group_snake_order, conv_mfma.h) instead of longest-first; 1200 frames too
cd $GRAFT_REPO_ROOT
S0="-DCG_C=256 -DCG_L=4992"; S1="-DCG_C=256 -DCG_L=9600"
T0="-DCG_CI=64 -DCG_MB=2 -DCG_NB=1 -DCG_WN=1 -DCG_KS=8 -DCG_WM=1"
bash tools/gpu/rb_diag.sh r04_diag18 "$S0 $T0" "$S0 -DRB_NEW=1 -DCG_NB=2" "$S0 -DRB_NEW=1 -DCG_NB=2 -DRB_ORDER=2" "$S1 $T0" "$S1 -DRB_NEW=1 -DCG_NB=2" "$S1 -DRB_NEW=1 -DCG_NB=2 -DRB_ORDER=2" > /dev/null
grep -E "^##|member k|L x8|1 stream|2 stream|4 stream" gpurun_out/r04_diag18/rb_diag.log
}

ab_snake() {
# the 256-channel stage promoted to the continuous-stream tile with the snake dispatch order (run_group / group_snake_order)
# against the 64 x 32 k-split tile (MI355TTS_NO_GROUP_PROMOTE=1): kernel stats, bench, parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab16; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --no-steady-state"
for v in old new; do
  e=0; [ $v = old ] && e=1
  MI355TTS_NO_GROUP_PROMOTE=$e timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t --output-format csv -- $B > $O/t_$v.log 2>&1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "rb_|conv_group|pair" $f | cut -c1-150
  rm -rf $O/t_$v
done
B2="python bench.py --no-cpu-baseline --no-config3 --no-config5 --no-half-mode"
for i in 1 2; do
  MI355TTS_NO_GROUP_PROMOTE=1 timeout 300 $B2 > $O/old_$i.json 2> $O/old_$i.err
  timeout 300 $B2 > $O/new_$i.json 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab16/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); p = j["profile_ms_per_step"]
    print(f.split("/")[-1][:-5], "value %.1f" % j["value"], "steady %.1f" % j["steady_state"]["utterances_per_sec"], "lat %.3f" % j["latency_ms_single_stream"], "frac %.4f" % j["roofline"]["frac"],
          "resblock %.3f" % p["conv_mfma.hifigan_resblock"], "config4 %.0f (%.3f ms single)" % (j["config4"]["utterances_per_sec"], j["config4"]["latency_ms_single_stream"]))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
}

diag_row_stride() {
# does the power-of-two factor of the activation row stride matter?  stage-1 geometry at 617 / 624 / 640 / 656 frames (strides of
# 2^8 x 617, 2^12 x 39, 2^15 x 5, 2^12 x 41 bytes) in the harness, and the vocoder alone at those frame counts
cd $GRAFT_REPO_ROOT
S="-DCG_C=128 -DRB_NEW=1"
bash tools/gpu/rb_diag.sh r04_diag19 "$S -DCG_L=39488" "$S -DCG_L=39936" "$S -DCG_L=40960" "$S -DCG_L=41984" > /dev/null
grep -E "^##|L x8|1 stream|2 stream" gpurun_out/r04_diag19/rb_diag.log
for f in 617 624 640 656; do echo "== frames $f"; PYTHONPATH=. timeout 300 python tools/ab_inproc.py --frames $f --rounds 4 larynx_amd/libmi355tts.so 2>&1 | grep wall; done
}

ab_bf16_snake() {
# split-bf16 mode: the 256-channel stage on 128 x 64 tiles dealt as a snake (BF_E, promote_group_plans) against the 8-wave
# k-split 128 x 128 tile (MI355TTS_NO_GROUP_PROMOTE=1): kernel stats, the half-mode bench line, parity.  NOT adopted
# (profiles/r04_ab17.txt): the BF_E tile and its promotion were taken out again — this function records how it was measured
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_ab17; mkdir -p $O
B="python bench.py --precision bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --no-steady-state"
for v in old new; do
  e=0; [ $v = old ] && e=1
  MI355TTS_NO_GROUP_PROMOTE=$e timeout 300 rocprofv3 --kernel-trace --stats -d $O/t_$v -o t --output-format csv -- $B > $O/t_$v.log 2>&1
  f=$(find $O/t_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep -E "bf16" $f | cut -c1-150
  rm -rf $O/t_$v
done
B2="python bench.py --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-steady-state"
for i in 1 2; do
  MI355TTS_NO_GROUP_PROMOTE=1 timeout 300 $B2 > $O/old_$i.json 2> $O/old_$i.err
  timeout 300 $B2 > $O/new_$i.json 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab17/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1]); h = j["half_mode"]
    print(f.split("/")[-1][:-5], "value %.1f" % j["value"], "half: %.1f utt/s" % h["utterances_per_sec"], "lat %.3f ms" % h["latency_ms_single_stream"], "class %.3f ms" % h["resblock_class_ms_per_step"], "frac %.3f" % h["roofline"]["frac"])
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16" 2>&1 | grep -E "passed|failed|Error" | tail -3
}

ab_promote_rule() {
# when does the 128-row tile + snake beat the k-split tile on the 256-channel stage?  the vocoder alone at several utterance
# lengths with the promotion forced (MI355TTS_PROMOTE_MAX_IMBALANCE=100), off (=0) and by the rule (busiest CU <= 1.2 x mean);
# then config 3 (256 utterances of 60 ... 200 ids) with the rule and without promotion
cd $GRAFT_REPO_ROOT
for f in 360 440 520 560 600 680 700 760 840 1040 1100; do
  for v in 100 0 1.2; do
    echo -n "frames $f max_imbalance $v: "; MI355TTS_PROMOTE_MAX_IMBALANCE=$v PYTHONPATH=. timeout 300 python tools/ab_inproc.py --frames $f --rounds 3 --calls 10 larynx_amd/libmi355tts.so 2>&1 | grep wall | sed "s/libmi355tts.so *//"
  done
done
O=gpurun_out/r04_ab18; mkdir -p $O
B2="python bench.py --no-cpu-baseline --no-config4 --no-config5 --no-half-mode --no-steady-state"
for i in 1 2; do
  MI355TTS_NO_GROUP_PROMOTE=1 timeout 400 $B2 > $O/old_$i.json 2> $O/old_$i.err
  timeout 400 $B2 > $O/new_$i.json 2> $O/new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab18/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1][:-5], "value %.1f" % j["value"], "config3 %.1f utt/s" % j["config3"]["utterances_per_sec"])
PY
}

pmc_by_stage() {
# SQ counters of the headline command with rb_group_kernel's dispatches split by launch geometry: the 256-channel stage (<= 1024
# workgroups) and the 128-channel stage (> 1024), promotion on and off (off: conv_group_kernel runs the 256-channel stage)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pmc_stage; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 --no-config4 --no-config5 --no-half-mode --concurrency 1 --repeats 1 --no-steady-state"
for v in new old; do
  e=0; [ $v = old ] && e=1
  MI355TTS_NO_GROUP_PROMOTE=$e timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq_$v -o sq --output-format csv -- $B > $O/sq_$v.log 2>&1
  f=$(find $O/sq_$v -name "*counter_collection.csv" | head -1)
  python tools/pmc_reduce.py $f --split-workgroups rb_group_kernel 1024 | grep -E "^kernel|rb_group|conv_group_kernel<11, 7, 3, 64, 2" > $O/by_stage_$v.csv
  rm -rf $O/sq_$v
done
python - <<'PY'
import csv, collections
for v in ("old", "new"):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(f"gpurun_out/r04_pmc_stage/by_stage_{v}.csv")):
        d[r["kernel"]][r["counter"]] = (int(r["dispatches"]), float(r["sum"]))
    for k, c in d.items():
        n, busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]; gui = c["GRBM_GUI_ACTIVE"][1]
        print(v, k, "dispatches", n, "MFMA busy %.3f" % (busy / (gui / 8 * 1024)), "active cycles per launch %.0f (per XCD)" % (gui / 8 / n),
              "WAIT_INST_ANY/WAVE_CYCLES %.2f" % (c["SQ_WAIT_INST_ANY"][1] / c["SQ_WAVE_CYCLES"][1]))
PY
}

"$@"
