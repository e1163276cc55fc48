"""One `Engine` per process per GPU: owns the C-ABI context, the resident models
and the thin numpy marshalling around the inference entry points."""
from __future__ import annotations

import ctypes as C
import json
import threading
import typing
import weakref

import numpy as np

from . import ffi
from .hparams import GlowHParams, HifiGanHParams
from .weights import build_blob


class MelBatch:
    """Device-resident mel batch handed from GlowTTS to HiFi-GAN without a host
    round trip (the reference crosses host<->runtime twice per sentence,
    `larynx/__init__.py:231-256`)."""

    def __init__(self, engine: "Engine", handle: int):
        self._engine = engine
        self._h = C.c_void_p(handle)
        engine._live_mels.add(self)
        lib = engine.lib
        self.batch = ffi.check(lib, lib.mi355tts_mel_batch(self._h))
        self.channels = ffi.check(lib, lib.mi355tts_mel_channels(self._h))
        self.max_frames = ffi.check(lib, lib.mi355tts_mel_max_frames(self._h))
        fr = (C.c_int32 * self.batch)()
        ffi.check(lib, lib.mi355tts_mel_frames(self._h, fr))
        self.frames = np.array(fr[:], dtype=np.int32)

    @property
    def handle(self) -> C.c_void_p:
        if self._h is None:
            raise ValueError("mel batch already freed")
        return self._h

    def numpy(self, which: str = "raw") -> np.ndarray:
        """[B, M, max_frames] float32; `which` is "raw" (GlowTTS output) or
        "vocoder" (after the AudioSettings transforms)."""
        out = np.zeros((self.batch, self.channels, self.max_frames), np.float32)
        if self.max_frames:
            lib = self._engine.lib
            ffi.check(lib, lib.mi355tts_mel_copy(self.handle, 0 if which == "raw" else 1, out.ctypes.data, self.max_frames))
        return out

    @property
    def shape(self):
        return (self.batch, self.channels, self.max_frames)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy("raw")
        return a if dtype is None else a.astype(dtype)

    def free(self):
        """Return the device blocks to the engine's pool.  A mel must not outlive its
        context: `Engine.close()` frees whatever is still alive first."""
        if self._h is not None:
            h, self._h = self._h, None
            if self._engine._ctx is not None:
                self._engine.lib.mi355tts_mel_free(h)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, device: int = 0, library_path=None):
        self.lib = ffi.load_library(library_path)
        self.device = device
        h = C.c_void_p()
        ffi.check(self.lib, self.lib.mi355tts_create(device, C.byref(h)))
        self._ctx = h
        self._lock = threading.Lock()
        self._hops: typing.Dict[int, int] = {}
        self._live_mels: "weakref.WeakSet[MelBatch]" = weakref.WeakSet()

    def close(self):
        if self._ctx is not None:
            for m in list(self._live_mels):  # their device blocks belong to this context
                m.free()
            self.lib.mi355tts_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -----------------------------------------------------------
    def glow_blob(self, hp: GlowHParams, state_dict) -> np.ndarray:
        return build_blob(ffi.manifest(self.lib, ffi.glow_hparams_c(hp)), state_dict)

    def hifigan_blob(self, hp: HifiGanHParams, state_dict) -> np.ndarray:
        return build_blob(ffi.manifest(self.lib, ffi.hifigan_hparams_c(hp)), state_dict)

    def load_glow(self, hp: GlowHParams, state_dict=None, blob=None, device_ptr: int = 0) -> int:
        """Load from a reference state-dict, a prebuilt host blob, or a device
        pointer holding the blob (the receive side of the RCCL weight broadcast)."""
        hp_c = ffi.glow_hparams_c(hp)
        return self._load(self.lib.mi355tts_load_glow, hp_c, state_dict, blob, device_ptr)

    def load_hifigan(self, hp: HifiGanHParams, state_dict=None, blob=None, device_ptr: int = 0) -> int:
        hp_c = ffi.hifigan_hparams_c(hp)
        return self._load(self.lib.mi355tts_load_hifigan, hp_c, state_dict, blob, device_ptr)

    def _load(self, fn, hp_c, state_dict, blob, device_ptr) -> int:
        man = ffi.manifest(self.lib, hp_c)
        total = sum(n for _, n in man)
        model = C.c_int()
        if device_ptr:
            ffi.check(self.lib, fn(self._ctx, C.byref(hp_c), C.c_void_p(device_ptr), total, 1, C.byref(model)))
        else:
            if blob is None:
                blob = build_blob(man, state_dict)
            blob = np.ascontiguousarray(blob, np.float32)
            if blob.size != total:
                raise ValueError(f"blob has {blob.size} floats, model needs {total}")
            ffi.check(self.lib, fn(self._ctx, C.byref(hp_c), blob.ctypes.data, total, 0, C.byref(model)))
        return int(model.value)

    def broadcast_weights(self, nccl_comm: int, root: int, device_ptr: int, numel: int, rccl_library: typing.Optional[str] = None):
        """`mi355tts_broadcast_weights`: ncclBroadcast of a device-resident weight blob over the caller's RCCL
        communicator (an `ncclComm_t` as an integer), in place."""
        lib_name = rccl_library.encode() if rccl_library else None
        ffi.check(self.lib, self.lib.mi355tts_broadcast_weights(self._ctx, C.c_void_p(nccl_comm), int(root), C.c_void_p(device_ptr),
                                                                int(numel), lib_name))

    def set_precision(self, model: int, precision: int):
        """`ffi.PRECISION_F32` (exact, default), `ffi.PRECISION_F16` (the reference's `half` switch: native fp16 vocoder; on a
        GlowTTS model the decoder's WaveNets in fp16) or `ffi.PRECISION_BF16X3` (split-bf16, f32-class accuracy; vocoder only).
        Returns the library's status: 0, or `ffi.PRECISION_NOOP` when the request has no effect on this GlowTTS model (it keeps
        computing in f32).  Raises when fp16 does not cover the vocoder's geometry."""
        return ffi.check(self.lib, self.lib.mi355tts_model_set_precision(self._ctx, int(model), int(precision)))

    def unload(self, model: int):
        ffi.check(self.lib, self.lib.mi355tts_unload(self._ctx, model))

    # ---- inference -----------------------------------------------------------
    def glow_infer(
        self,
        model: int,
        ids: typing.Union[np.ndarray, typing.Sequence[np.ndarray]],
        noise_scale: float = 0.667,
        length_scale: float = 1.0,
        noise: typing.Optional[np.ndarray] = None,
        seed: int = 0,
        audio_settings=None,
        row_seeds: typing.Optional[typing.Sequence[int]] = None,
        speaker_ids: typing.Union[None, int, typing.Sequence[int]] = None,
    ) -> MelBatch:
        """`ids`: one int64 vector [P] or a list of them (variable length batch).
        `speaker_ids`: a multi-speaker voice's speaker per row (one int = every row; the reference's `speaker_id`
        setting, larynx/glow_tts.py:116-130) — required there, an error for a single-speaker voice.
        `noise`: optional [B, M, >=F] (or [M, >=F] for B=1) standing in for the
        reference's `torch.randn_like` draw.  Without it the device generator draws: row b from
        the stream `row_seeds[b]` (default `seed + b`: successive BATCHED calls must advance `seed` by the batch size)
        — the field a batch-1 call with that seed draws."""
        rows = [np.asarray(ids, np.int64)] if isinstance(ids, np.ndarray) and ids.ndim == 1 else [np.asarray(r, np.int64) for r in ids]
        if isinstance(ids, np.ndarray) and ids.ndim == 2:
            rows = [np.asarray(r, np.int64) for r in ids]
        B = len(rows)
        lens = np.array([len(r) for r in rows], np.int32)
        ld = int(lens.max()) if B else 0
        packed = np.zeros((B, max(ld, 1)), np.int64)
        for b, r in enumerate(rows):
            packed[b, : len(r)] = r
        nz_ptr, nz_ld = None, 0
        if noise is not None:
            noise = np.ascontiguousarray(noise, np.float32)
            if noise.ndim == 2:
                noise = noise[None]
            if noise.shape[0] != B:
                raise ValueError("noise batch mismatch")
            nz_ptr, nz_ld = noise.ctypes.data, noise.shape[2]
        a = ffi.audio_settings_c(audio_settings) if audio_settings is not None else None
        out = C.c_void_p()
        if row_seeds is not None and (noise is not None or len(row_seeds) != B):
            raise ValueError("row_seeds: one seed per row, and no explicit noise")
        if speaker_ids is not None:
            spk = self._speaker_array(speaker_ids, B)
            rs = None if row_seeds is None else np.array([int(x) & (2 ** 64 - 1) for x in row_seeds], np.uint64)
            ffi.check(
                self.lib,
                self.lib.mi355tts_glow_infer_speakers(
                    self._ctx, model, packed.ctypes.data, lens.ctypes.data_as(C.POINTER(C.c_int32)), B, packed.shape[1],
                    float(noise_scale), float(length_scale), nz_ptr, nz_ld, int(seed) & (2 ** 64 - 1),
                    rs.ctypes.data_as(C.POINTER(C.c_uint64)) if rs is not None else None,
                    spk.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(a) if a is not None else None, 0, C.byref(out),
                ),
            )
            return MelBatch(self, out.value)
        if row_seeds is not None:
            rs = np.array([int(x) & (2 ** 64 - 1) for x in row_seeds], np.uint64)
            ffi.check(
                self.lib,
                self.lib.mi355tts_glow_infer_rows(
                    self._ctx, model, packed.ctypes.data, lens.ctypes.data_as(C.POINTER(C.c_int32)), B, packed.shape[1],
                    float(noise_scale), float(length_scale), rs.ctypes.data_as(C.POINTER(C.c_uint64)),
                    C.byref(a) if a is not None else None, 0, C.byref(out),
                ),
            )
            return MelBatch(self, out.value)
        ffi.check(
            self.lib,
            self.lib.mi355tts_glow_infer(
                self._ctx, model, packed.ctypes.data, lens.ctypes.data_as(C.POINTER(C.c_int32)), B, packed.shape[1],
                float(noise_scale), float(length_scale), nz_ptr, nz_ld, int(seed) & (2 ** 64 - 1),
                C.byref(a) if a is not None else None, 0, C.byref(out),
            ),
        )
        return MelBatch(self, out.value)

    @staticmethod
    def _speaker_array(speaker_ids, B: int) -> np.ndarray:
        spk = np.full(B, int(speaker_ids), np.int32) if np.isscalar(speaker_ids) else np.asarray(list(speaker_ids), np.int32)
        if spk.shape != (B,):
            raise ValueError("speaker_ids: one speaker per row")
        return np.ascontiguousarray(spk)

    def glow_infer_raw(self, model, ids_ptr, lens, ids_ld, noise_scale, length_scale, noise_ptr=None, noise_ld=0,
                       seed=0, audio_settings=None, flags=0) -> MelBatch:
        """Pointer-level entry (ids/noise may be device memory with flags=ffi.IN_DEVICE)."""
        lens = np.ascontiguousarray(lens, np.int32)
        a = ffi.audio_settings_c(audio_settings) if audio_settings is not None else None
        out = C.c_void_p()
        ffi.check(
            self.lib,
            self.lib.mi355tts_glow_infer(
                self._ctx, model, ids_ptr, lens.ctypes.data_as(C.POINTER(C.c_int32)), len(lens), ids_ld,
                float(noise_scale), float(length_scale), noise_ptr, noise_ld, int(seed) & (2 ** 64 - 1),
                C.byref(a) if a is not None else None, flags, C.byref(out),
            ),
        )
        return MelBatch(self, out.value)

    def hifigan_infer_raw(self, vocoder, mel: MelBatch, f32_ptr, i16_ptr, wav_ld, flags=0, denoiser_strength=0.0,
                          pad_before=0, pad_after=0):
        ffi.check(self.lib, self.lib.mi355tts_hifigan_infer_padded(self._ctx, vocoder, mel.handle, float(denoiser_strength),
                                                                  f32_ptr, i16_ptr, wav_ld, flags, int(pad_before), int(pad_after)))

    def synthesize_raw(self, glow, vocoder, ids_ptr, lens, ids_ld, noise_scale, length_scale, f32_ptr, i16_ptr, wav_ld,
                       noise_ptr=None, noise_ld=0, seed=0, audio_settings=None, denoiser_strength=0.0, pad_before=0,
                       pad_after=0, flags=0) -> np.ndarray:
        """Pointer-level fused call (`mi355tts_synthesize`); returns the per-row mel frame counts."""
        lens = np.ascontiguousarray(lens, np.int32)
        a = ffi.audio_settings_c(audio_settings) if audio_settings is not None else None
        frames = np.zeros(len(lens), np.int32)
        ffi.check(
            self.lib,
            self.lib.mi355tts_synthesize(
                self._ctx, glow, vocoder, ids_ptr, lens.ctypes.data_as(C.POINTER(C.c_int32)), len(lens), ids_ld,
                float(noise_scale), float(length_scale), noise_ptr, noise_ld, int(seed) & (2 ** 64 - 1),
                C.byref(a) if a is not None else None, float(denoiser_strength), int(pad_before), int(pad_after),
                frames.ctypes.data_as(C.POINTER(C.c_int32)), f32_ptr, i16_ptr, wav_ld, flags,
            ),
        )
        return frames

    def synthesize(self, glow: int, vocoder: int, ids, noise_scale: float = 0.667, length_scale: float = 1.0,
                   noise: typing.Optional[np.ndarray] = None, seed: int = 0, audio_settings=None,
                   denoiser_strength: float = 0.0, pad_before: int = 0, pad_after: int = 0, want_float: bool = False,
                   frames_per_id_guess: float = 8.0, speaker_ids: typing.Union[None, int, typing.Sequence[int]] = None):
        """ids -> (frames [B], wav_f32 [B, n] or None, wav_i16 [B, n]) through ONE fused call
        (`mi355tts_synthesize`): row b holds pad_before zeros, frames[b]*hop samples, then zeros.
        The frame count is data dependent; the output buffer is sized from a guess and the call
        is repeated once with the exact size in the rare case the guess was too small."""
        rows = [np.asarray(ids, np.int64)] if isinstance(ids, np.ndarray) and ids.ndim == 1 else [np.asarray(r, np.int64) for r in ids]
        B = len(rows)
        lens = np.array([len(r) for r in rows], np.int32)
        ld = int(lens.max())
        packed = np.zeros((B, ld), np.int64)
        for b, r in enumerate(rows):
            packed[b, : len(r)] = r
        nz_ptr, nz_ld = None, 0
        if noise is not None:
            noise = np.ascontiguousarray(noise, np.float32)
            if noise.ndim == 2:
                noise = noise[None]
            nz_ptr, nz_ld = noise.ctypes.data, noise.shape[2]
        hop = self.hop(vocoder)
        pads = int(pad_before) + int(pad_after)
        cap = int(ld * frames_per_id_guess * max(length_scale, 0.05)) * hop + pads
        lens_c = lens.ctypes.data_as(C.POINTER(C.c_int32))
        a = ffi.audio_settings_c(audio_settings) if audio_settings is not None else None
        frames = np.zeros(B, np.int32)
        spk = self._speaker_array(speaker_ids, B) if speaker_ids is not None else None
        for attempt in range(2):
            f32 = np.empty((B, cap), np.float32) if want_float else None
            i16 = np.empty((B, cap), np.int16)
            head = (self._ctx, glow, vocoder, packed.ctypes.data, lens_c, B, ld, float(noise_scale), float(length_scale), nz_ptr,
                    nz_ld, int(seed) & (2 ** 64 - 1))
            tail = (C.byref(a) if a is not None else None, float(denoiser_strength), int(pad_before), int(pad_after),
                    frames.ctypes.data_as(C.POINTER(C.c_int32)), ffi.ptr(f32), i16.ctypes.data, cap, 0)
            if spk is not None:
                rc = self.lib.mi355tts_synthesize_speakers(*head, spk.ctypes.data_as(C.POINTER(C.c_int32)), *tail)
            else:
                rc = self.lib.mi355tts_synthesize(*head, *tail)
            n = int(frames.max()) * hop + pads
            if rc == -4 and attempt == 0 and n > cap:  # MI355TTS_ERR_TOO_SMALL: frames_out holds the real counts
                cap = n
                continue
            ffi.check(self.lib, rc)
            return frames, (f32[:, :n] if f32 is not None else None), i16[:, :n]
        raise AssertionError("unreachable")

    def reserve(self, workers: int, glow: int = 0, vocoder: int = 0, max_batch: int = 1, max_ids: int = 256,
                max_frames: int = 2048, denoiser: bool = False, max_pad_samples: int = 0):
        """Pre-create per-call workers and size every workspace (`mi355tts_reserve`)."""
        ffi.check(self.lib, self.lib.mi355tts_reserve(self._ctx, int(workers), int(glow), int(vocoder), int(max_batch), int(max_ids),
                                                     int(max_frames), 1 if denoiser else 0, int(max_pad_samples)))

    def ensure_workers(self, n: int):
        """At least `n` per-call workers exist and their streams' hardware queues are known (`mi355tts_reserve` without models: it
        creates the workers, measures which of their streams share a hardware queue — calls are then spread evenly over the queues —
        and leaves the workspaces to grow on first use).  Idempotent; the hosts of a sentence thread pool call it with the pool's
        size + 1 (larynx/__init__.py:146-157: one `_sentence_task` per pool thread)."""
        n = max(1, min(int(n), 64))
        if n > getattr(self, "_ensured_workers", 0):
            self.reserve(n, 0, 0, max_batch=1, max_ids=1, max_frames=1)
            self._ensured_workers = n

    def worker_queue_groups(self) -> typing.List[int]:
        """The hardware-queue group of every worker, in creation order (`mi355tts_worker_queue_groups`; -1 = not probed)."""
        import ctypes

        buf = (ctypes.c_int32 * 256)()
        n = self.lib.mi355tts_worker_queue_groups(self._ctx, buf, 256)
        if n < 0:
            ffi.check(self.lib, n)
        return [int(buf[i]) for i in range(min(n, 256))]

    def mel_from_numpy(self, mel: np.ndarray, frames=None, audio_settings=None) -> MelBatch:
        mel = np.ascontiguousarray(mel, np.float32)
        if mel.ndim == 2:
            mel = mel[None]
        B, M, ld = mel.shape
        fr = np.full(B, ld, np.int32) if frames is None else np.asarray(frames, np.int32)
        a = ffi.audio_settings_c(audio_settings) if audio_settings is not None else None
        out = C.c_void_p()
        ffi.check(
            self.lib,
            self.lib.mi355tts_mel_from_buffer(
                self._ctx, mel.ctypes.data, fr.ctypes.data_as(C.POINTER(C.c_int32)), B, M, ld,
                C.byref(a) if a is not None else None, 0, C.byref(out),
            ),
        )
        return MelBatch(self, out.value)

    def mel_from_device(self, device_ptr: int, frames, channels: int, ld: int, audio_settings=None) -> MelBatch:
        """Wrap a mel that already lives in device memory ([B][channels][ld] fp32, e.g. a
        torch tensor's `data_ptr()`); `frames` gives each row's valid length."""
        fr = np.ascontiguousarray(frames, np.int32)
        a = ffi.audio_settings_c(audio_settings) if audio_settings is not None else None
        out = C.c_void_p()
        ffi.check(
            self.lib,
            self.lib.mi355tts_mel_from_buffer(
                self._ctx, C.c_void_p(int(device_ptr)), fr.ctypes.data_as(C.POINTER(C.c_int32)), len(fr), int(channels), int(ld),
                C.byref(a) if a is not None else None, ffi.IN_DEVICE, C.byref(out),
            ),
        )
        return MelBatch(self, out.value)

    def hop(self, vocoder: int) -> int:
        if vocoder not in self._hops:
            self._hops[vocoder] = ffi.check(self.lib, self.lib.mi355tts_hifigan_hop(self._ctx, vocoder))
        return self._hops[vocoder]

    def hifigan_infer(self, vocoder: int, mel: MelBatch, want_float: bool = True, want_int16: bool = True,
                      denoiser_strength: float = 0.0, pad_before: int = 0, pad_after: int = 0):
        """Returns (wav_f32 [B, N] or None, wav_i16 [B, N] or None), N = pad_before + max_frames*hop + pad_after;
        row b holds pad_before zeros, frames[b]*hop samples, then zeros (the SSML pauses of
        `larynx/__init__.py:277-283`, written by the int16 kernel instead of np.pad)."""
        n = mel.max_frames * self.hop(vocoder) + int(pad_before) + int(pad_after)
        f32 = np.empty((mel.batch, n), np.float32) if want_float else None
        i16 = np.empty((mel.batch, n), np.int16) if want_int16 else None
        ffi.check(
            self.lib,
            self.lib.mi355tts_hifigan_infer_padded(self._ctx, vocoder, mel.handle, float(denoiser_strength), ffi.ptr(f32),
                                                   ffi.ptr(i16), n, 0, int(pad_before), int(pad_after)),
        )
        return f32, i16

    # ---- single operators ------------------------------------------------------
    def conv1d(self, x, w, bias=None, dilation=1, in_slope=1.0, out_act=0, lens=None) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        B, Cin, L = x.shape
        Cout, _, K = w.shape
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        y = np.zeros((B, Cout, L), np.float32)
        ln = None if lens is None else np.ascontiguousarray(lens, np.int32)
        ffi.check(
            self.lib,
            self.lib.mi355tts_op_conv1d(
                self._ctx, x.ctypes.data, B, Cin, L, None if ln is None else ln.ctypes.data_as(C.POINTER(C.c_int32)),
                w.ctypes.data, ffi.ptr(b), Cout, K, int(dilation), float(in_slope), int(out_act), y.ctypes.data,
            ),
        )
        return y

    def gauss_noise(self, seed: int, B: int, channels: int, T: int) -> np.ndarray:
        """[B, channels, T] draws of the device noise generator (what `glow_infer` adds when no
        explicit noise tensor is given)."""
        out = np.empty((B, channels, T), np.float32)
        ffi.check(self.lib, self.lib.mi355tts_op_gauss_noise(self._ctx, int(seed) & (2 ** 64 - 1), B, channels, T, out.ctypes.data))
        return out

    def denoise(self, wav: np.ndarray, bias_spec: np.ndarray, strength: float) -> np.ndarray:
        wav = np.ascontiguousarray(wav, np.float32)
        if wav.ndim == 1:
            wav = wav[None]
        bias = np.ascontiguousarray(bias_spec, np.float32)
        out = np.zeros_like(wav)
        ffi.check(self.lib, self.lib.mi355tts_op_denoise(self._ctx, wav.ctypes.data, wav.shape[0], wav.shape[1], bias.ctypes.data,
                                                        float(strength), out.ctypes.data))
        return out

    def conv_transpose1d(self, x, w, bias, stride, in_slope=1.0) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        B, Cin, L = x.shape
        _, Cout, K = w.shape
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        y = np.zeros((B, Cout, L * stride), np.float32)
        ffi.check(
            self.lib,
            self.lib.mi355tts_op_conv_transpose1d(
                self._ctx, x.ctypes.data, B, Cin, L, w.ctypes.data, ffi.ptr(b), Cout, K, int(stride), float(in_slope), y.ctypes.data
            ),
        )
        return y

    def bench_conv1d(self, B, Cin, Cout, K, dilation, L, tile_shape=-1, iters=20) -> float:
        """Average ms per launch of the conv kernel on device-resident random data."""
        ms = C.c_float()
        ffi.check(self.lib, self.lib.mi355tts_bench_conv1d(self._ctx, B, Cin, Cout, K, dilation, L, tile_shape, iters, C.byref(ms)))
        return float(ms.value)

    # ---- measurement -------------------------------------------------------------
    def set_profiling(self, on: bool):
        ffi.check(self.lib, self.lib.mi355tts_set_profiling(self._ctx, 1 if on else 0))

    def set_option(self, name: str, value: int):
        ffi.check(self.lib, self.lib.mi355tts_set_option(self._ctx, name.encode("ascii"), int(value)))

    def coalesce_stats(self):
        """(passes, rows) of the shared GlowTTS passes since the context was created (see include/mi355tts.h)."""
        a, b = C.c_int64(0), C.c_int64(0)
        ffi.check(self.lib, self.lib.mi355tts_coalesce_stats(self._ctx, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def profile_reset(self):
        ffi.check(self.lib, self.lib.mi355tts_profile_reset(self._ctx))

    def profile_event_overhead_us(self, pairs: int = 64) -> float:
        """Median elapsed time of an EMPTY event pair on an idle stream (`mi355tts_profile_event_overhead`)."""
        us = C.c_double(0.0)
        ffi.check(self.lib, self.lib.mi355tts_profile_event_overhead(self._ctx, int(pairs), C.byref(us)))
        return float(us.value)

    def profile(self) -> dict:
        buf = C.create_string_buffer(4096)
        ffi.check(self.lib, self.lib.mi355tts_profile_json(self._ctx, buf, 4096))
        return json.loads(buf.value.decode("ascii"))

    def get_call_coalesce_default(self) -> int:
        """The library's built-in default of option `call_coalesce` (`mi355tts_call_coalesce_default`)."""
        return int(self.lib.mi355tts_call_coalesce_default())

    def profile_kernels(self) -> dict:
        """The profiled launches per kernel NAME and sub-key (`mi355tts_profile_kernels_json`): {class: {"name/sub": {...}}}."""
        buf = C.create_string_buffer(32768)
        ffi.check(self.lib, self.lib.mi355tts_profile_kernels_json(self._ctx, buf, 32768))
        return json.loads(buf.value.decode("ascii"))

    def dispatch_selfcheck(self) -> dict:
        """The one-off check of the dispatch-order assumption (`mi355tts_dispatch_selfcheck`): runs it if it has not run."""
        st, a, b = C.c_int(0), C.c_float(0.0), C.c_float(0.0)
        ffi.check(self.lib, self.lib.mi355tts_dispatch_selfcheck(self._ctx, C.byref(st), C.byref(a), C.byref(b)))
        return {"state": {0: "not run", 1: "snake order kept", 2: "snake order switched off", 3: "skipped", 4: "running"}[int(st.value)],
                "plain_us": float(a.value), "snake_us": float(b.value)}

    def kernel_counts(self) -> dict:
        """Launches per kernel name since the last `profile_reset` (`mi355tts_kernel_counts_json`; always counted)."""
        buf = C.create_string_buffer(4096)
        ffi.check(self.lib, self.lib.mi355tts_kernel_counts_json(self._ctx, buf, 4096))
        return json.loads(buf.value.decode("ascii"))
