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
