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
