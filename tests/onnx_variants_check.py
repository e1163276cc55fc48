"""Helper of tests/test_onnx_ingestion.py (own process: it imports the reference tree): export the reference's modules
with the given exporter settings, ingest the files with larynx_amd.onnx_weights, print the worst blob difference.
Usage: onnx_variants_check.py OPSET FOLD KEEP_INITIALIZERS EMU_LIBRARY OUT_DIR"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    opset, fold, keep = int(sys.argv[1]), bool(int(sys.argv[2])), bool(int(sys.argv[3]))
    emu, out = sys.argv[4], Path(sys.argv[5])
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils

    from larynx_amd import ffi, synthetic
    from larynx_amd.onnx_weights import state_dict_from_onnx
    from larynx_amd.weights import build_blob
    from oracle.make_golden import build_ref_glow, build_ref_hifigan, import_reference
    from oracle.make_onnx_fixture import GLOW, VOC

    onnx_proto_utils._add_onnxscript_fn = lambda proto, custom_opsets: proto  # needs the absent `onnx` package; a no-op here
    gm, hm, hc, _ = import_reference()
    gsd = synthetic.make_glow_state_dict(GLOW, seed=31)
    vsd = synthetic.make_hifigan_state_dict(VOC, seed=32)
    lib = ffi.load_library(emu)

    def worst(path, man, sd, n_split=4):
        got = state_dict_from_onnx(path, [n for n, _ in man], n_split=n_split)
        return float(np.abs(build_blob(man, got) - build_blob(man, sd)).max())

    gen = build_ref_hifigan(hm, hc, VOC, vsd)
    with torch.no_grad():
        torch.onnx.export(gen, torch.randn(1, 80, 20), str(out / "v.onnx"), opset_version=opset, do_constant_folding=fold,
                          keep_initializers_as_inputs=keep, input_names=["mel"], output_names=["audio"],
                          dynamic_axes={"mel": {2: "frames"}, "audio": {2: "samples"}}, dynamo=False)
    res = {"hifigan": worst(out / "v.onnx", ffi.manifest(lib, ffi.hifigan_hparams_c(VOC)), vsd)}

    model = build_ref_glow(gm, GLOW, gsd)
    for p in model.parameters():
        p.requires_grad_(False)
    for f in model.decoder.flows:
        if hasattr(f, "weight_inv"):
            f.weight_inv = f.weight_inv.detach()

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, text, lengths, scales):
            (mel, *_), _, _ = self.m(text, lengths, noise_scale=scales[0], length_scale=scales[1], g=None)
            return mel

    text = torch.randint(1, GLOW.num_symbols, (1, 17))
    with torch.no_grad():
        torch.onnx.export(Wrap(model), (text, torch.LongTensor([17]), torch.FloatTensor([0.667, 1.0])), str(out / "g.onnx"),
                          opset_version=opset, do_constant_folding=fold, keep_initializers_as_inputs=keep,
                          input_names=["input", "input_lengths", "scales"], output_names=["output"],
                          dynamic_axes={"input": {0: "batch", 1: "phonemes"}, "input_lengths": {0: "batch"},
                                        "output": {0: "batch", 2: "frames"}}, dynamo=False)
    res["glow"] = worst(out / "g.onnx", ffi.manifest(lib, ffi.glow_hparams_c(GLOW)), gsd, GLOW.n_split)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
