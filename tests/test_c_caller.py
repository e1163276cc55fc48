"""A plain C99 program (tests/c_abi/caller.c) drives include/mi355tts.h end to end —
manifest, load, batched glow_infer, mel accessors, hifigan_infer, error returns — and
its checksums must equal the same calls made through larynx_amd.ffi: the ABI really is
plain C, usable without Python (the cgo/JNI/ctypes binding case of INTEGRATION.md)."""
import ctypes as C
import re
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from larynx_amd import ffi
from larynx_amd import hparams as HP

REPO = Path(__file__).resolve().parent.parent
SRC = REPO / "tests" / "c_abi" / "caller.c"

GLOW = HP.GlowHParams(num_symbols=20, hidden_channels=32, filter_channels=48, filter_channels_dp=24, kernel_size=3,
                      n_blocks_dec=2, n_layers_enc=1, n_heads=2, dilation_rate=1, kernel_size_dec=5, n_block_layers=2,
                      n_sqz=2, prenet=True, window_size=4, n_split=4, mel_channels=8)
VOC = HP.HifiGanHParams(resblock="1", upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4), upsample_initial_channel=16,
                        resblock_kernel_sizes=(3, 7), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5)), num_mels=8)


def python_side(lib_path):
    lib = ffi.load_library(lib_path)
    ctx = C.c_void_p()
    ffi.check(lib, lib.mi355tts_create(0, C.byref(ctx)))
    g_c, v_c = ffi.glow_hparams_c(GLOW), ffi.hifigan_hparams_c(VOC)
    state = 12345
    blobs = []
    for hp_c in (g_c, v_c):
        man = ffi.manifest(lib, hp_c)
        # continue the generator across the two blobs exactly as the C program does
        parts = []
        for name, n in man:
            vals = np.empty(n, np.float32)
            for k in range(n):
                state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
                v = np.float32(0.08) * (np.float32(state >> 8) / np.float32(8388608.0) - np.float32(1.0))
                if "gamma" in name:
                    v = v + np.float32(1.0)
                if "_inv" in name and n == 16 and k % 5 == 0:
                    v = v + np.float32(1.0)
                vals[k] = v
            parts.append(vals)
        blobs.append(np.concatenate(parts))
    glow, voc = C.c_int(-1), C.c_int(-1)
    fp = C.POINTER(C.c_float)
    ffi.check(lib, lib.mi355tts_load_glow(ctx, C.byref(g_c), blobs[0].ctypes.data_as(fp), blobs[0].size, 0, C.byref(glow)))
    ffi.check(lib, lib.mi355tts_load_hifigan(ctx, C.byref(v_c), blobs[1].ctypes.data_as(fp), blobs[1].size, 0, C.byref(voc)))
    ids = np.array([[3, 5, 9, 3, 12, 7, 3, 2, 0], [3, 4, 18, 6, 2, 0, 0, 0, 0]], np.int64)
    lens = np.array([8, 5], np.int32)
    audio = ffi.AudioSettingsC(1, 1, 1, 1, 1, -100.0, 1.0, 20.0, 1.0)
    mel = C.c_void_p()
    ffi.check(lib, lib.mi355tts_glow_infer(ctx, glow, ids.ctypes.data, lens.ctypes.data_as(C.POINTER(C.c_int32)), 2, 9, 0.0, 1.0,
                                           None, 0, 1, C.byref(audio), 0, C.byref(mel)))
    frames = np.zeros(2, np.int32)
    ffi.check(lib, lib.mi355tts_mel_frames(mel, frames.ctypes.data_as(C.POINTER(C.c_int32))))
    M, F = lib.mi355tts_mel_channels(mel), lib.mi355tts_mel_max_frames(mel)
    hop = lib.mi355tts_hifigan_hop(ctx, voc)
    raw = np.zeros((2, M, F), np.float32)
    ffi.check(lib, lib.mi355tts_mel_copy(mel, 0, raw.ctypes.data_as(fp), F))
    wav = np.zeros((2, F * hop), np.float32)
    pcm = np.zeros((2, F * hop), np.int16)
    ffi.check(lib, lib.mi355tts_hifigan_infer(ctx, voc, mel, 0.0, wav.ctypes.data_as(fp), pcm.ctypes.data_as(C.POINTER(C.c_int16)), F * hop, 0))
    lib.mi355tts_mel_free(mel)
    lib.mi355tts_destroy(ctx)
    i = np.arange(F * hop)
    return (int(frames[0]), int(frames[1]), float(raw.astype(np.float64).sum()),
            float((wav.astype(np.float64) * ((i % 7) + 1)).sum()), int((pcm.astype(np.int64) * ((i % 5) + 1)).sum()))


def run_c(lib_path, tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = tmp_path / "caller"
    lib_path = Path(lib_path).resolve()
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", f"-I{REPO / 'include'}", str(SRC), "-o", str(exe),
                    str(lib_path), f"-Wl,-rpath,{lib_path.parent}"], check=True)
    out = subprocess.run([str(exe), "0"], check=True, capture_output=True, text=True, timeout=600).stdout
    m = re.match(r"frames (\d+) (\d+) mel_sum (\S+) wav_sum (\S+) pcm_sum (-?\d+)", out)
    assert m, out
    return int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(5))


def compare(lib_path, tmp_path):
    c = run_c(lib_path, tmp_path)
    p = python_side(lib_path)
    assert c[:2] == p[:2] and c[0] > 0 and c[1] > 0
    assert c[2] == pytest.approx(p[2], rel=1e-6, abs=1e-6)
    assert c[3] == pytest.approx(p[3], rel=1e-6, abs=1e-6)
    assert c[4] == p[4] and c[4] != 0


def test_c_caller_matches_python_binding_on_emulator(emu_library, tmp_path):
    compare(emu_library, tmp_path)


@pytest.mark.gpu
def test_c_caller_matches_python_binding_on_gpu(tmp_path):
    compare(ffi.DEFAULT_LIBRARY, tmp_path)
