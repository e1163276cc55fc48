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
