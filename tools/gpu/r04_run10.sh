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
