"""Checkpoint -> the flat fp32 blob `mi355tts_load_glow/_hifigan` ingest.

What the reference does at load time (`larynx/glow_tts.py:66-95`,
`larynx/hifi_gan.py:71-100`): read `checkpoint["model"]` / `dict["generator"]`,
`remove_weight_norm()`, `decoder.store_inverse()`.  Here the same folding happens
once on the host in numpy; tensor ORDER comes from the library's own manifest
(`mi355tts_*_manifest`), so this file never hard-codes the blob layout.
"""
from __future__ import annotations

import typing
from pathlib import Path

import numpy as np

StateDict = typing.Mapping[str, typing.Any]


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):  # torch tensor, without importing torch here
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """`w = g * v / ||v||` over all dims but 0 (torch.nn.utils.weight_norm, dim=0),
    i.e. what `remove_weight_norm` bakes in (hifi_gan/models.py:204-211,
    glow_tts/layers.py:164-170; `CouplingBlock.start` stays weight-normed at run
    time in the reference, attentions.py:96-98 — folded here as well)."""
    v64 = v.astype(np.float64)
    norm = np.sqrt((v64 ** 2).reshape(v.shape[0], -1).sum(axis=1)).reshape(g.shape)
    return (v64 * (g.astype(np.float64) / norm)).astype(np.float32)


def resolve_tensor(sd: StateDict, name: str) -> np.ndarray:
    """Look `name` (a manifest entry) up in a reference state-dict."""
    if name.endswith(".weight_inv"):
        if name in sd:  # an ONNX export carries the stored inverse itself (onnx_weights.py)
            return _np(sd[name]).astype(np.float32)
        # InvConvNear.store_inverse: torch.inverse(weight.float()) (layers.py:274-275)
        w = _np(sd[name[: -len("_inv")]]).astype(np.float32)
        return np.linalg.inv(w).astype(np.float32)
    if name in sd:
        return _np(sd[name]).astype(np.float32)
    if name.endswith(".weight"):
        base = name[: -len(".weight")]
        if base + ".weight_g" in sd:
            return fold_weight_norm(_np(sd[base + ".weight_g"]), _np(sd[base + ".weight_v"]))
    raise KeyError(f"checkpoint has no tensor for '{name}'")


def build_blob(manifest: typing.Sequence[typing.Tuple[str, int]], sd: StateDict) -> np.ndarray:
    total = sum(n for _, n in manifest)
    blob = np.empty(total, np.float32)
    pos = 0
    for name, n in manifest:
        t = resolve_tensor(sd, name)
        if t.size != n:
            raise ValueError(f"'{name}': checkpoint has {t.size} elements, model needs {n}")
        blob[pos : pos + n] = t.reshape(-1)
        pos += n
    return blob


def load_state_dict(path: typing.Union[str, Path], key: str, manifest_names: typing.Optional[typing.Sequence[str]] = None,
                    n_split: int = 4) -> StateDict:
    """Read a reference `generator.pth` (torch pickle, `{"model": sd}` for GlowTTS,
    `{"generator": sd}` for HiFi-GAN), this project's `.npz` of the same keys, or a released voice's
    `generator.onnx` (its initializers, see onnx_weights.py; needs the library's manifest names)."""
    path = Path(path)
    if path.suffix == ".onnx":
        from .onnx_weights import state_dict_from_onnx

        if manifest_names is None:
            raise ValueError("reading an ONNX file needs the manifest's tensor names")
        return state_dict_from_onnx(path, manifest_names, n_split=n_split)
    if path.suffix == ".npz":
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    import torch  # only needed for .pth ingestion

    obj = torch.load(path, map_location="cpu", weights_only=True)
    if key in obj:
        obj = obj[key]
    return obj
