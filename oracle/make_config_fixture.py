#!/usr/bin/env python
"""TEST INFRASTRUCTURE — collect the hyper-parameter blocks of every voice and vocoder
config the reference ships (`local/<lang>/<voice>-glow_tts/config.json`,
`local/hifi_gan/*/config.json`) into one small fixture, so the CPU suite can check that the
library accepts all of them without `/root/reference` being present.
Run in the build container:  python oracle/make_config_fixture.py [/root/reference]"""
import json
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "reference_configs.json"

voices = {}
for cfg in sorted(REF.glob("local/*/*-glow_tts/config.json")):
    d = json.loads(cfg.read_text())
    name = f"{cfg.parent.parent.name}/{cfg.parent.name}"
    voices[name] = {"model": d["model"], "audio": d.get("audio", {})}
vocoders = {}
for cfg in sorted(REF.glob("local/hifi_gan/*/config.json")):
    d = json.loads(cfg.read_text())
    d.pop("dist_config", None)  # inert training leftovers
    vocoders[cfg.parent.name] = d
OUT.write_text(json.dumps({"source": "rhasspy/larynx local/*/config.json", "voices": voices, "vocoders": vocoders}, indent=0, sort_keys=True))
print(f"{len(voices)} voices, {len(vocoders)} vocoders -> {OUT} ({OUT.stat().st_size} bytes)")
