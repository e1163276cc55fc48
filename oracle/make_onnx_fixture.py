"""TEST INFRASTRUCTURE — export the reference's own modules to ONNX, the way its voices were produced.

Runs ONLY in the build container (needs /root/reference and torch).  Released Larynx voices are
`torch.onnx.export`s of `glow_tts.models.FlowGenerator` (after `decoder.store_inverse()`) and
`hifi_gan.models.Generator` (after `remove_weight_norm()`), with the input names the reference's ONNX
feed dicts use (`larynx/glow_tts.py:161-168`: input / input_lengths / scales; `larynx/hifi_gan.py:150`: mel).
This script does the same for shrunk hyper-parameters with seeded synthetic checkpoints and writes
`tests/golden/onnx/{glow,hifigan}/generator.onnx` + `config.json` + the source state-dicts as `.npz`,
so that `larynx_amd/onnx_weights.py` (initializer ingestion) is tested against
real exporter output everywhere.  The `onnx` package is not installed: the exporter's optional
onnxscript post-processing hook (which imports it) is bypassed — the serialized graph is unaffected.

Usage:  python -m oracle.make_onnx_fixture
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from larynx_amd import hparams as HP  # noqa: E402
from larynx_amd import synthetic  # noqa: E402
from oracle.make_golden import build_ref_glow, build_ref_hifigan, import_reference  # noqa: E402

OUT = REPO / "tests" / "golden" / "onnx"
# 80 mel channels: the reference's Generator hard-codes them (hifi_gan/models.py:153)
GLOW = HP.GlowHParams(num_symbols=46, hidden_channels=32, filter_channels=64, filter_channels_dp=40, n_blocks_dec=2,
                      n_layers_enc=2, n_block_layers=2, mel_channels=80)
VOC = HP.HifiGanHParams(upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4), upsample_initial_channel=32,
                        resblock_kernel_sizes=(3, 5), resblock_dilation_sizes=((1, 3, 5), (1, 2, 3)), num_mels=80)


def main():
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto  # needs the absent `onnx` package; a no-op here
    gm, hm, hc, _ = import_reference()
    gsd = synthetic.make_glow_state_dict(GLOW, seed=31)
    vsd = synthetic.make_hifigan_state_dict(VOC, seed=32)
    (OUT / "glow").mkdir(parents=True, exist_ok=True)
    (OUT / "hifigan").mkdir(parents=True, exist_ok=True)

    gen = build_ref_hifigan(hm, hc, VOC, vsd)
    with torch.no_grad():
        torch.onnx.export(gen, torch.randn(1, 80, 20), str(OUT / "hifigan" / "generator.onnx"), opset_version=12,
                          input_names=["mel"], output_names=["audio"],
                          dynamic_axes={"mel": {2: "frames"}, "audio": {2: "samples"}}, dynamo=False)
    (OUT / "hifigan" / "config.json").write_text(json.dumps(VOC.to_config()))
    np.savez_compressed(OUT / "hifigan" / "state_dict.npz", **vsd)

    model = build_ref_glow(gm, GLOW, gsd)
    for p in model.parameters():
        p.requires_grad_(False)
    for f in model.decoder.flows:  # store_inverse() already ran (build_ref_glow); its result must be a plain constant for the tracer
        if hasattr(f, "weight_inv"):
            f.weight_inv = f.weight_inv.detach()

    class Wrap(torch.nn.Module):  # the three-input signature of the reference's ONNX session (larynx/glow_tts.py:161-168)
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, text, lengths, scales):
            (mel, *_), _, _ = self.m(text, lengths, noise_scale=scales[0], length_scale=scales[1], g=None)
            return mel

    text = torch.randint(1, GLOW.num_symbols, (1, 17))
    with torch.no_grad():
        torch.onnx.export(Wrap(model), (text, torch.LongTensor([17]), torch.FloatTensor([0.667, 1.0])),
                          str(OUT / "glow" / "generator.onnx"), opset_version=12,
                          input_names=["input", "input_lengths", "scales"], output_names=["output"],
                          dynamic_axes={"input": {0: "batch", 1: "phonemes"}, "input_lengths": {0: "batch"},
                                        "output": {0: "batch", 2: "frames"}}, dynamo=False)
    (OUT / "glow" / "config.json").write_text(json.dumps(GLOW.to_config()))
    np.savez_compressed(OUT / "glow" / "state_dict.npz", **gsd)
    for p in sorted(OUT.rglob("*")):
        if p.is_file():
            print(p.relative_to(REPO), p.stat().st_size)


if __name__ == "__main__":
    main()
