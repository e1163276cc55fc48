"""TEST INFRASTRUCTURE — build larynx_amd/csrc against the CPU HIP emulator."""
from __future__ import annotations

import hashlib
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
CSRC = REPO / "larynx_amd" / "csrc"
OUT_DIR = REPO / "build"


def _clangxx() -> str:
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "clang++"):
        if Path(c).is_file() or c == "clang++":
            return c
    return "clang++"


def build_emu() -> Path:
    srcs = [CSRC / "mi355tts.hip", HERE / "hipemu_runtime.cpp"]
    deps = sorted(list(CSRC.glob("*")) + list((HERE / "include" / "hip").glob("*")) + [HERE / "hipemu_runtime.cpp", REPO / "include" / "mi355tts.h"])
    h = hashlib.sha1()
    for d in deps:
        if d.is_file():
            h.update(d.read_bytes())
    OUT_DIR.mkdir(exist_ok=True)
    out = OUT_DIR / f"libmi355tts_emu_{h.hexdigest()[:12]}.so"
    if out.is_file():
        return out
    cmd = [
        _clangxx(), "-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-Wno-psabi", "-Wno-unused-value",
        f"-I{HERE / 'include'}", *map(str, srcs), "-o", str(out), "-lpthread",
    ]
    subprocess.run(cmd, check=True, cwd=str(REPO))
    for old in OUT_DIR.glob("libmi355tts_emu_*.so"):  # builds of earlier source states
        if old != out:
            try:
                old.unlink()
            except OSError:
                pass
    return out


if __name__ == "__main__":
    print(build_emu())
