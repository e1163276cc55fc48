"""Build the HIP library in-tree: `hipcc --offload-arch=gfx950` on csrc/mi355tts.hip
-> larynx_amd/libmi355tts.so (cross-compiles without a GPU).  `python -m larynx_amd.build`."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
REPO = PKG.parent
CSRC = PKG / "csrc"
OUT = PKG / "libmi355tts.so"
STAMP = PKG / ".libmi355tts.stamp"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and Path(c).is_file():
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def source_hash() -> str:
    h = hashlib.sha1()
    for p in sorted(list(CSRC.glob("*")) + [REPO / "include" / "mi355tts.h"]):
        if p.is_file():
            h.update(p.name.encode())
            h.update(p.read_bytes())
    return h.hexdigest()


def resources_hash() -> str:
    """The key tests/test_kernel_resources.py files the compiler's resource remarks under."""
    h = hashlib.sha1()
    for f in sorted(CSRC.glob("*")) + [REPO / "include" / "mi355tts.h"]:
        h.update(f.read_bytes())
    return h.hexdigest()[:12]


def build(force: bool = False, verbose: bool = False) -> Path:
    digest = source_hash()
    if not force and OUT.is_file() and STAMP.is_file() and STAMP.read_text().strip() == digest:
        return OUT
    # -Rpass-analysis=kernel-resource-usage: the compiler's per-kernel VGPR / scratch / LDS / occupancy remarks, kept under
    # build/ (git-ignored) keyed by the source state — tests/test_kernel_resources.py reads them instead of compiling the
    # device code a second time (two to three minutes)
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
        "-Rpass-analysis=kernel-resource-usage", str(CSRC / "mi355tts.hip"), "-o", str(OUT),
    ]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, cwd=str(REPO), stderr=subprocess.PIPE, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stderr[-8000:])
        raise subprocess.CalledProcessError(proc.returncode, cmd)
    try:
        cache_dir = REPO / "build"
        cache_dir.mkdir(exist_ok=True)
        for old in cache_dir.glob("kernel_resources_*.txt"):
            old.unlink()
        (cache_dir / f"kernel_resources_{resources_hash()}.txt").write_text(proc.stderr)
    except OSError:
        pass
    STAMP.write_text(digest)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
