"""Process-wide engines (one per GPU) and voice-directory loading."""
from __future__ import annotations

import json
import threading
import typing
from pathlib import Path

from .engine import Engine

_lock = threading.Lock()
_engines: typing.Dict[typing.Tuple[int, typing.Optional[str]], Engine] = {}


def get_engine(device: int = 0, library_path=None) -> Engine:
    """The reference keeps global model caches (`larynx/__init__.py:290,412`);
    here the per-GPU context is the cached object."""
    key = (device, str(library_path) if library_path else None)
    with _lock:
        if key not in _engines:
            _engines[key] = Engine(device=device, library_path=library_path)
        return _engines[key]


def find_checkpoint(model_path: Path) -> Path:
    """What `valid_voice_dir` (`larynx/utils.py:203-209`) accepts — `generator.pth` (the reference's torch
    checkpoint) or `generator.onnx` (what released voices ship; its initializers are read, onnx_weights.py) —
    or this project's `generator.npz`."""
    for name in ("generator.npz", "generator.pth", "generator.onnx"):
        p = Path(model_path) / name
        if p.is_file():
            return p
    onnx = sorted(Path(model_path).glob("*.onnx")) if Path(model_path).is_dir() else []
    if onnx:
        return onnx[0]
    raise FileNotFoundError(f"{model_path}: no generator.pth / generator.onnx / generator.npz")


def read_config(model_path: Path) -> dict:
    with open(Path(model_path) / "config.json", "r", encoding="utf-8") as f:
        return json.load(f)
