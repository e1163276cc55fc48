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
