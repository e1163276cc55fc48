"""`generator.onnx` ingestion: the dependency-free ONNX reader and the graph
matching that recovers the tensors the exporter folds or renames, on files this image's `torch.onnx.export`
produced from the REFERENCE'S OWN modules (`oracle/make_onnx_fixture.py`; committed under tests/golden/onnx).
The blob built from the ONNX file must equal the blob built from the checkpoint the export started from, and
a voice directory that holds only `generator.onnx` must synthesise the same audio."""
import json
import shutil
from pathlib import Path

import numpy as np
import pytest

import larynx_amd
from larynx_amd import ffi
from larynx_amd import hparams as HP
from larynx_amd.constants import TextToSpeechType, VocoderType
from larynx_amd.onnx_reader import read_onnx
from larynx_amd.onnx_weights import state_dict_from_onnx
from larynx_amd.weights import build_blob

FIX = Path(__file__).resolve().parent / "golden" / "onnx"


def _cfg(which):
    cfg = json.loads((FIX / which / "config.json").read_text())
    hp = HP.GlowHParams.from_config(cfg) if which == "glow" else HP.HifiGanHParams.from_config(cfg)
    with np.load(FIX / which / "state_dict.npz") as z:
        sd = {k: z[k] for k in z.files}
    return hp, sd


def test_reader_parses_the_exporters_files():
    g = read_onnx(FIX / "hifigan" / "generator.onnx")
    assert g.inputs == ["mel"] and g.outputs == ["audio"]
    assert sum(n.op_type == "Conv" for n in g.nodes) == 26 and sum(n.op_type == "ConvTranspose" for n in g.nodes) == 2
    assert g.initializers["conv_pre.weight"].shape == (32, 80, 7) and g.initializers["conv_pre.weight"].dtype == np.float32
    conv = next(n for n in g.nodes if n.op_type == "Conv")
    assert conv.attrs["kernel_shape"] == [7] and conv.attrs["pads"] == [3, 3] and conv.inputs[0] == "mel"
    g2 = read_onnx(FIX / "glow" / "generator.onnx")
    assert g2.inputs == ["input", "input_lengths", "scales"] and g2.outputs == ["output"]  # larynx/glow_tts.py:161-168
    assert len(g2.nodes) > 1000 and any(n.op_type == "Identity" for n in g2.nodes) is not None


@pytest.mark.parametrize("which", ["hifigan", "glow"])
def test_onnx_blob_equals_checkpoint_blob(emu_library_path, which):
    hp, sd = _cfg(which)
    lib = ffi.load_library(emu_library_path)
    man = ffi.manifest(lib, ffi.hifigan_hparams_c(hp) if which == "hifigan" else ffi.glow_hparams_c(hp))
    names = [n for n, _ in man]
    from_onnx = state_dict_from_onnx(FIX / which / "generator.onnx", names, n_split=getattr(hp, "n_split", 4))
    a = build_blob(man, from_onnx)
    b = build_blob(man, sd)
    assert a.shape == b.shape
    pos, worst = 0, {}
    for name, n in man:
        d = float(np.abs(a[pos : pos + n] - b[pos : pos + n]).max())
        if d:
            worst[name] = d
        pos += n
    # tensors the checkpoint stores under the same name come back bit for bit; what the reference folds before
    # exporting (remove_weight_norm in float32 vs our float64 fold, torch.inverse, exp(-logs)) to f32 round-off
    exact = [n for n, _ in man if n in sd and not n.endswith(".logs")]
    assert exact and all(n not in worst for n in exact), {k: v for k, v in worst.items() if k in exact}
    assert all(v < 2e-6 for v in worst.values()), worst


def test_voice_directories_with_only_onnx_files(emu_library_path, tmp_path):
    """What a released voice looks like to `valid_voice_dir` (larynx/utils.py:203-209): config.json + generator.onnx."""
    ghp, gsd = _cfg("glow")
    vhp, vsd = _cfg("hifigan")
    dirs = {}
    for kind in ("onnx", "npz"):
        gdir, vdir = tmp_path / f"{kind}-glow_tts", tmp_path / f"{kind}_hifi_gan"
        gdir.mkdir()
        vdir.mkdir()
        shutil.copy(FIX / "glow" / "config.json", gdir / "config.json")
        shutil.copy(FIX / "hifigan" / "config.json", vdir / "config.json")
        if kind == "onnx":
            shutil.copy(FIX / "glow" / "generator.onnx", gdir / "generator.onnx")
            shutil.copy(FIX / "hifigan" / "generator.onnx", vdir / "generator.onnx")
        else:
            np.savez(gdir / "generator.npz", **gsd)
            np.savez(vdir / "generator.npz", **vsd)
        dirs[kind] = (gdir, vdir)
    ids = np.array([3, 8, 4, 14, 3, 35, 3, 26, 4, 34, 22, 3, 2], np.int64)
    audio = {}
    for kind, (gdir, vdir) in dirs.items():
        tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, library_path=emu_library_path)
        voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, library_path=emu_library_path)
        audio[kind] = voc.mels_to_audio(tts.phonemes_to_mels(ids, {"noise_scale": 0.0}))
    assert audio["onnx"].shape == audio["npz"].shape and audio["onnx"].size > 0
    assert np.abs(audio["onnx"].astype(np.int32) - audio["npz"].astype(np.int32)).max() <= 1


def test_unrecognised_layouts_fail_loudly(tmp_path):
    hp, _ = _cfg("hifigan")
    with pytest.raises(KeyError):
        state_dict_from_onnx(FIX / "hifigan" / "generator.onnx", ["conv_pre.weight", "no.such.tensor"])
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x08\x03")  # a ModelProto with an ir_version and no graph
    with pytest.raises(ValueError):
        read_onnx(bad)


@pytest.mark.parametrize("opset,fold,keep", [(11, True, False), (13, False, True), (17, True, False)])
def test_exporter_variants_from_the_reference_modules(emu_library_path, tmp_path, opset, fold, keep):
    """Other exporter settings than the committed fixture's (opset 12, folding on): older / newer opsets, constant
    folding off, initializers kept as graph inputs — exported from the reference's own modules by
    `tests/onnx_variants_check.py` in a process of its own (it imports the reference tree; skipped where /root/reference
    does not exist, e.g. on the GPU box).  All 20 combinations of opset {11, 12, 13, 14, 17} x folding x
    keep_initializers_as_inputs were checked once with this code; three of them run in the suite."""
    if not Path("/root/reference/glow_tts/models.py").is_file():
        pytest.skip("needs the reference checkout")
    import subprocess
    import sys

    p = subprocess.run([sys.executable, str(Path(__file__).with_name("onnx_variants_check.py")), str(opset), str(int(fold)), str(int(keep)),
                        str(emu_library_path), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    worst = json.loads(p.stdout.strip().splitlines()[-1])
    assert worst["hifigan"] < 2e-6 and worst["glow"] < 2e-6, worst


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _with_extra_initializer(src, dst, tensor: bytes):
    """Copy an ONNX file, appending one TensorProto to graph.initializer (ModelProto field 7 -> GraphProto field 5)."""
    from larynx_amd.onnx_reader import _fields

    out = bytearray()
    for f, w, v in _fields(src.read_bytes()):
        if f == 7 and w == 2:
            v = bytes(v) + b"\x2a" + _varint(len(tensor)) + tensor
        if w == 2:
            out += _varint((f << 3) | 2) + _varint(len(v)) + bytes(v)
        elif w == 0:
            out += _varint((f << 3) | 0) + _varint(v)
        else:
            raise AssertionError("unexpected wire type in a ModelProto")
    dst.write_bytes(bytes(out))


def test_an_unrelated_unreadable_initializer_does_not_block_a_voice(tmp_path):
    """A file may carry a tensor this reader cannot take (data in an external file, odd element count).  Only tensors the
    weight loader needs turn that into an error — and then the error says why."""
    name = b"extra.lookup_table"
    ext = b"\x08\x02" + b"\x10\x01" + b"\x42" + _varint(len(name)) + name + b"\x70\x01"  # dims [2], FLOAT, name, data_location = EXTERNAL
    f = tmp_path / "generator.onnx"
    _with_extra_initializer(FIX / "hifigan" / "generator.onnx", f, ext)
    g = read_onnx(f)
    assert "extra.lookup_table" in g.unreadable and "extra.lookup_table" not in g.initializers
    ref = state_dict_from_onnx(FIX / "hifigan" / "generator.onnx", ["conv_pre.weight", "conv_pre.bias"])
    got = state_dict_from_onnx(f, ["conv_pre.weight", "conv_pre.bias"])
    assert sorted(got) == sorted(ref) and all(np.array_equal(got[k], ref[k]) for k in ref)
    with pytest.raises(ValueError, match="external file"):
        state_dict_from_onnx(f, ["conv_pre.weight", "extra.lookup_table"])
