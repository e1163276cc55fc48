"""ctypes binding of `include/mi355tts.h` — the only way Python reaches the HIP path.

The shared library is the hipcc-built `larynx_amd/libmi355tts.so` (built by
`__graft_entry__.build()` / `python -m larynx_amd.build`).  There is no CPU
fallback: if the library is missing or fails to load, `load_library()` raises.
ctypes releases the GIL for the duration of every foreign call, which is what
lets Larynx's `ThreadPoolExecutor` (`larynx/__init__.py:146`) overlap sentences.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import typing
from pathlib import Path

import numpy as np

PACKAGE_DIR = Path(__file__).resolve().parent
DEFAULT_LIBRARY = PACKAGE_DIR / "libmi355tts.so"

MAX_STAGES = 8
IN_DEVICE = 1
OUT_DEVICE = 2
ABI_VERSION = 2  # MI355TTS_ABI_VERSION of include/mi355tts.h
PRECISION_F32 = 0
PRECISION_BF16X3 = 1
PRECISION_BF16 = 2
PRECISION_F16 = 3  # the reference's `.half()`: native fp16 vocoder (csrc/conv_f16.h), fp16 decoder WaveNets of GlowTTS (csrc/wn_f16.h)
PRECISION_NOOP = 1  # mi355tts_model_set_precision's return on a GlowTTS model the fp16 kernel does not cover: accepted, no effect


class GlowHParamsC(C.Structure):
    _fields_ = [
        (n, C.c_int32)
        for n in (
            "num_symbols", "hidden_channels", "filter_channels", "filter_channels_dp",
            "kernel_size", "n_blocks_dec", "n_layers_enc", "n_heads",
            "dilation_rate", "kernel_size_dec", "n_block_layers", "n_sqz",
            "prenet", "window_size", "n_split", "mel_channels",
            "prenet_kernel_size", "prenet_layers", "n_speakers", "gin_channels",
        )
    ]


class HifiGanHParamsC(C.Structure):
    _fields_ = [
        ("resblock_type", C.c_int32),
        ("num_upsamples", C.c_int32),
        ("upsample_rates", C.c_int32 * MAX_STAGES),
        ("upsample_kernel_sizes", C.c_int32 * MAX_STAGES),
        ("upsample_initial_channel", C.c_int32),
        ("num_kernels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * MAX_STAGES),
        ("num_dilations", C.c_int32),
        ("resblock_dilations", (C.c_int32 * MAX_STAGES) * MAX_STAGES),
        ("num_mels", C.c_int32),
    ]


class AudioSettingsC(C.Structure):
    _fields_ = [
        ("signal_norm", C.c_int32),
        ("symmetric_norm", C.c_int32),
        ("clip_norm", C.c_int32),
        ("convert_db_to_amp", C.c_int32),
        ("do_dynamic_range_compression", C.c_int32),
        ("min_level_db", C.c_float),
        ("max_norm", C.c_float),
        ("ref_level_db", C.c_float),
        ("spec_gain", C.c_float),
    ]


class Mi355ttsError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mi355tts error {code}: {message}")
        self.code = code


# every exported symbol of include/mi355tts.h: name -> (restype, argtypes)
_VP = C.c_void_p
_SIGNATURES: typing.Dict[str, typing.Tuple[typing.Any, typing.List[typing.Any]]] = {
    "mi355tts_abi_version": (C.c_int, []),
    "mi355tts_last_error": (C.c_char_p, []),
    "mi355tts_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "mi355tts_destroy": (None, [_VP]),
    "mi355tts_glow_manifest": (C.c_int, [C.POINTER(GlowHParamsC), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]),
    "mi355tts_hifigan_manifest": (C.c_int, [C.POINTER(HifiGanHParamsC), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]),
    "mi355tts_load_glow": (C.c_int, [_VP, C.POINTER(GlowHParamsC), _VP, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
    "mi355tts_load_hifigan": (C.c_int, [_VP, C.POINTER(HifiGanHParamsC), _VP, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
    "mi355tts_unload": (C.c_int, [_VP, C.c_int]),
    "mi355tts_model_set_precision": (C.c_int, [_VP, C.c_int, C.c_int]),
    "mi355tts_broadcast_weights": (C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int64, C.c_char_p]),
    "mi355tts_glow_infer_rows": (
        C.c_int,
        [_VP, C.c_int, _VP, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_uint64),
         C.POINTER(AudioSettingsC), C.c_uint32, C.POINTER(_VP)],
    ),
    "mi355tts_glow_infer": (
        C.c_int,
        [_VP, C.c_int, _VP, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_float, _VP, C.c_int, C.c_uint64,
         C.POINTER(AudioSettingsC), C.c_uint32, C.POINTER(_VP)],
    ),
    "mi355tts_glow_infer_speakers": (
        C.c_int,
        [_VP, C.c_int, _VP, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_float, _VP, C.c_int, C.c_uint64,
         C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(AudioSettingsC), C.c_uint32, C.POINTER(_VP)],
    ),
    "mi355tts_mel_batch": (C.c_int, [_VP]),
    "mi355tts_mel_channels": (C.c_int, [_VP]),
    "mi355tts_mel_max_frames": (C.c_int, [_VP]),
    "mi355tts_mel_frames": (C.c_int, [_VP, C.POINTER(C.c_int32)]),
    "mi355tts_mel_copy": (C.c_int, [_VP, C.c_int, _VP, C.c_int]),
    "mi355tts_mel_free": (None, [_VP]),
    "mi355tts_mel_from_buffer": (
        C.c_int,
        [_VP, _VP, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(AudioSettingsC), C.c_uint32, C.POINTER(_VP)],
    ),
    "mi355tts_hifigan_hop": (C.c_int, [_VP, C.c_int]),
    "mi355tts_hifigan_infer": (C.c_int, [_VP, C.c_int, _VP, C.c_float, _VP, _VP, C.c_int64, C.c_uint32]),
    "mi355tts_hifigan_infer_padded": (C.c_int, [_VP, C.c_int, _VP, C.c_float, _VP, _VP, C.c_int64, C.c_uint32, C.c_int32, C.c_int32]),
    "mi355tts_synthesize": (
        C.c_int,
        [_VP, C.c_int, C.c_int, _VP, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_float, _VP, C.c_int, C.c_uint64,
         C.POINTER(AudioSettingsC), C.c_float, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _VP, _VP, C.c_int64, C.c_uint32],
    ),
    "mi355tts_synthesize_speakers": (
        C.c_int,
        [_VP, C.c_int, C.c_int, _VP, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_float, C.c_float, _VP, C.c_int, C.c_uint64,
         C.POINTER(C.c_int32), C.POINTER(AudioSettingsC), C.c_float, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _VP, _VP,
         C.c_int64, C.c_uint32],
    ),
    "mi355tts_reserve": (C.c_int, [_VP] + [C.c_int] * 8),
    "mi355tts_worker_queue_groups": (C.c_int, [_VP, C.POINTER(C.c_int32), C.c_int]),
    "mi355tts_op_gauss_noise": (C.c_int, [_VP, C.c_uint64, C.c_int, C.c_int, C.c_int, _VP]),
    "mi355tts_op_denoise": (C.c_int, [_VP, _VP, C.c_int, C.c_int64, _VP, C.c_float, _VP]),
    "mi355tts_op_conv1d": (
        C.c_int,
        [_VP, _VP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _VP],
    ),
    "mi355tts_op_conv_transpose1d": (
        C.c_int,
        [_VP, _VP, C.c_int, C.c_int, C.c_int, _VP, _VP, C.c_int, C.c_int, C.c_int, C.c_float, _VP],
    ),
    "mi355tts_bench_conv1d": (C.c_int, [_VP] + [C.c_int] * 8 + [C.POINTER(C.c_float)]),
    "mi355tts_set_profiling": (C.c_int, [_VP, C.c_int]),
    "mi355tts_set_option": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "mi355tts_coalesce_stats": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mi355tts_profile_reset": (C.c_int, [_VP]),
    "mi355tts_profile_json": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "mi355tts_call_coalesce_default": (C.c_int, []),
    "mi355tts_profile_kernels_json": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "mi355tts_kernel_counts_json": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "mi355tts_dispatch_selfcheck": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mi355tts_profile_event_overhead": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_double)]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib_lock = threading.Lock()
_libs: typing.Dict[str, C.CDLL] = {}


def load_library(path: typing.Union[str, os.PathLike, None] = None) -> C.CDLL:
    """dlopen the C-ABI library and type every entry point.  Raises (never falls
    back) when the library is absent — build it with `python -m larynx_amd.build`."""
    p = Path(path) if path is not None else DEFAULT_LIBRARY
    key = str(p.resolve())
    with _lib_lock:
        if key in _libs:
            return _libs[key]
        if not p.is_file():
            raise FileNotFoundError(
                f"{p} not found: the HIP library has not been built "
                "(run `python -m larynx_amd.build`; hipcc --offload-arch=gfx950 is required)"
            )
        lib = C.CDLL(key)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the header and the .so disagree
            fn.restype = res
            fn.argtypes = args
        ver = lib.mi355tts_abi_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"{p}: ABI version {ver}, expected {ABI_VERSION}")
        _libs[key] = lib
        return lib


def check(lib: C.CDLL, rc: int) -> int:
    if rc < 0:
        msg = lib.mi355tts_last_error()
        raise Mi355ttsError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc


def glow_hparams_c(hp) -> GlowHParamsC:
    c = GlowHParamsC()
    for name, _ in GlowHParamsC._fields_:
        setattr(c, name, int(getattr(hp, name)))
    return c


def hifigan_hparams_c(hp) -> HifiGanHParamsC:
    c = HifiGanHParamsC()
    c.resblock_type = int(hp.resblock)
    c.num_upsamples = len(hp.upsample_rates)
    if len(hp.upsample_rates) > MAX_STAGES or len(hp.resblock_kernel_sizes) > MAX_STAGES:
        raise ValueError("too many HiFi-GAN stages")
    for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
        c.upsample_rates[i] = int(u)
        c.upsample_kernel_sizes[i] = int(k)
    c.upsample_initial_channel = int(hp.upsample_initial_channel)
    c.num_kernels = len(hp.resblock_kernel_sizes)
    nd = {len(d) for d in hp.resblock_dilation_sizes}
    if len(nd) != 1:
        raise ValueError("resblocks must share one dilation count")
    c.num_dilations = nd.pop()
    for j, (k, dil) in enumerate(zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes)):
        c.resblock_kernel_sizes[j] = int(k)
        for d, v in enumerate(dil):
            c.resblock_dilations[j][d] = int(v)
    c.num_mels = int(hp.num_mels)
    return c


def audio_settings_c(s) -> AudioSettingsC:
    c = AudioSettingsC()
    c.signal_norm = int(bool(s.signal_norm))
    c.symmetric_norm = int(bool(s.symmetric_norm))
    c.clip_norm = int(bool(s.clip_norm))
    c.convert_db_to_amp = int(bool(s.convert_db_to_amp))
    c.do_dynamic_range_compression = int(bool(s.do_dynamic_range_compression))
    c.min_level_db = float(s.min_level_db)
    c.max_norm = float(s.max_norm)
    c.ref_level_db = float(s.ref_level_db)
    c.spec_gain = float(s.spec_gain)
    return c


def manifest(lib: C.CDLL, hp_c) -> typing.List[typing.Tuple[str, int]]:
    fn = lib.mi355tts_glow_manifest if isinstance(hp_c, GlowHParamsC) else lib.mi355tts_hifigan_manifest
    out = []
    buf = C.create_string_buffer(256)
    n = C.c_int64()
    i = 0
    while True:
        rc = check(lib, fn(C.byref(hp_c), i, buf, 256, C.byref(n)))
        if rc == 1:
            return out
        out.append((buf.value.decode("ascii"), int(n.value)))
        i += 1


def ptr(a: typing.Optional[np.ndarray]) -> typing.Optional[int]:
    return None if a is None else a.ctypes.data
