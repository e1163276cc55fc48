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
