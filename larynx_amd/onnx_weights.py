"""`generator.onnx` -> the reference state-dict tensors the HIP backend loads.

Released Larynx voices carry their weights only as ONNX initializers
(`larynx/utils.py:19-21,203-209`; loaded by `larynx/glow_tts.py:98-100`, `larynx/hifi_gan.py:103-105`).
The graphs were exported with `torch.onnx.export` from the in-tree modules after
`remove_weight_norm()` / `store_inverse()` (glow-tts-train / hifi-gan-train `export_onnx`), so:

* every plain `nn.Parameter` keeps its state-dict name as the initializer name (possibly behind
  a wrapper prefix such as `m.` or `generator.`) — all of HiFi-GAN, most of GlowTTS;
* tensors the exporter constant-folds lose their names and are recovered from the GRAPH:
  - a conv weight is the weight input of the `Conv` node whose bias input is the named `<layer>.bias`
    (`CouplingBlock.start`, still weight-normed at export time: the folded `g * v / |v|`);
  - `InvConvNear.weight_inv` (an attribute, not a parameter): the `[n_split, n_split, 1, 1]` weights of the
    4x4 `Conv` nodes in execution order (reverse flow order: last block first, glow_tts/models.py:195-206);
  - LayerNorm `gamma` / `beta` (`x * gamma.view(1,-1,1) + beta.view(1,-1,1)`, glow_tts/layers.py:24-27): the
    constants of the k-th `Mul -> Add` pair with `[1, C, 1]` constants, in execution order
    (prenet, encoder layers, duration predictor — the order of the library's manifest);
  - ActNorm reverse (`(x - bias) * exp(-logs)`, glow_tts/layers.py:192-194) when the exporter folded `-logs`
    or `exp(-logs)` into an anonymous constant: the k-th `Sub(x, bias) -> Mul` pair in execution order, whose
    other factor is a `[1, C, 1]` constant (`logs = -log(scale)`), `Exp(constant)` or `Exp(Neg(constant))`;
  `Identity` aliases (the de-duplication pass of newer exporters) are followed.

Validated against files produced by this image's `torch.onnx.export` from the reference's own modules
(`oracle/make_onnx_fixture.py`, `tests/test_onnx_ingestion.py`); no released voice is available offline.
"""
from __future__ import annotations

import typing
from pathlib import Path

import numpy as np

from .onnx_reader import OnnxGraph, read_onnx

_ROOTS = ("encoder.", "decoder.", "conv_pre.", "conv_post.", "ups.", "resblocks.")


def _strip_prefix(name: str) -> str:
    best = None
    for r in _ROOTS:
        i = name.find(r)
        if i >= 0 and (i == 0 or name[i - 1] == ".") and (best is None or i < best):
            best = i
    return name[best:] if best is not None else name


class _Graph:
    def __init__(self, g: OnnxGraph):
        self.g = g
        self.alias = {n.outputs[0]: n.inputs[0] for n in g.nodes if n.op_type == "Identity" and n.inputs and n.outputs}
        self.producer = {o: n for n in g.nodes for o in n.outputs}

    def root(self, name: str) -> str:
        seen = 0
        while name in self.alias and seen < 64:
            name = self.alias[name]
            seen += 1
        return name

    def const(self, name: str) -> typing.Optional[np.ndarray]:
        return self.g.initializers.get(self.root(name))

    def const_of(self, node, shape_pred) -> typing.Optional[typing.Tuple[int, np.ndarray]]:
        for k, i in enumerate(node.inputs):
            c = self.const(i)
            if c is not None and shape_pred(c.shape):
                return k, c
        return None


def state_dict_from_onnx(path: typing.Union[str, Path], manifest_names: typing.Sequence[str], n_split: int = 4) -> typing.Dict[str, np.ndarray]:
    """Tensors for every entry of `manifest_names` (the library's manifest: reference state-dict names after
    weight-norm folding, plus `...weight_inv`), read from the ONNX file at `path`."""
    G = _Graph(read_onnx(path))
    named: typing.Dict[str, np.ndarray] = {}
    for raw, arr in G.g.initializers.items():
        s = _strip_prefix(raw)
        if s != raw or any(raw.startswith(r) for r in _ROOTS):
            named[s] = np.asarray(arr)
    # conv weights anchored by their named bias
    for n in G.g.nodes:
        if n.op_type in ("Conv", "ConvTranspose") and len(n.inputs) >= 3:
            b = _strip_prefix(G.root(n.inputs[2]))
            w = G.const(n.inputs[1])
            if b.endswith(".bias") and w is not None:
                named.setdefault(b[: -len(".bias")] + ".weight", np.asarray(w))
    out: typing.Dict[str, np.ndarray] = {}
    missing: typing.List[str] = []
    for name in manifest_names:
        if name in named:
            out[name] = named[name]
        elif name.endswith(".weight") and name[:-7] + ".weight_g" in named and name[:-7] + ".weight_v" in named:
            out[name[:-7] + ".weight_g"] = named[name[:-7] + ".weight_g"]
            out[name[:-7] + ".weight_v"] = named[name[:-7] + ".weight_v"]
        elif name.endswith(".weight_inv") and name[: -len("_inv")] in named:
            out[name[: -len("_inv")]] = named[name[: -len("_inv")]]  # weights.resolve_tensor inverts it
        else:
            missing.append(name)
    if not missing:
        return out

    def is_c1(shape):  # [1, C, 1]
        return len(shape) == 3 and shape[0] == 1 and shape[2] == 1 and shape[1] > 1

    # ---- graph patterns, in execution order
    inv_weights = [np.asarray(G.const(n.inputs[1])) for n in G.g.nodes
                   if n.op_type == "Conv" and len(n.inputs) >= 2 and G.const(n.inputs[1]) is not None
                   and tuple(G.const(n.inputs[1]).shape) == (n_split, n_split, 1, 1)]
    ln_pairs: typing.List[typing.Tuple[np.ndarray, np.ndarray]] = []
    an_pairs: typing.List[typing.Tuple[np.ndarray, np.ndarray]] = []
    consumers: typing.Dict[str, typing.List] = {}
    for n in G.g.nodes:
        for i in n.inputs:
            consumers.setdefault(i, []).append(n)

    def next_through_alias(out_name):
        return consumers.get(out_name, [])

    for n in G.g.nodes:
        if n.op_type == "Mul":
            c = G.const_of(n, is_c1)
            if c is None:
                continue
            for m in next_through_alias(n.outputs[0]):
                if m.op_type == "Add":
                    d = G.const_of(m, is_c1)
                    if d is not None and d[1].shape == c[1].shape:
                        ln_pairs.append((np.asarray(c[1]), np.asarray(d[1])))
                        break
        elif n.op_type == "Sub":
            c = G.const_of(n, is_c1)
            if c is None or c[0] != 1:  # x - bias: the constant is the subtrahend
                continue
            for m in next_through_alias(n.outputs[0]):
                if m.op_type != "Mul":
                    continue
                logs = None
                for i in m.inputs:
                    if i == n.outputs[0]:
                        continue
                    d = G.const(i)
                    if d is not None and is_c1(d.shape):  # exp(-logs) folded into one constant
                        logs = -np.log(np.asarray(d, np.float64))
                        break
                    p = G.producer.get(G.root(i))
                    if p is not None and p.op_type == "Exp":  # Exp(const = -logs)  or  Exp(Neg(logs))
                        e = G.const(p.inputs[0])
                        if e is not None and is_c1(e.shape):
                            logs = -np.asarray(e, np.float64)
                            break
                        q = G.producer.get(G.root(p.inputs[0]))
                        if q is not None and q.op_type == "Neg" and G.const(q.inputs[0]) is not None:
                            logs = np.asarray(G.const(q.inputs[0]), np.float64)
                            break
                if logs is not None and logs.shape == c[1].shape:
                    an_pairs.append((np.asarray(c[1]), logs.astype(np.float32)))
                    break

    def anchor_size(norm_name: str) -> typing.Optional[int]:
        """Channels the folded tensor `norm_name` must have: the rows of the named conv its layer follows (a structural
        anchor on top of the execution-order matching: an extra or displaced [1, C, 1] Mul -> Add pair of another width
        fails here instead of loading as garbage)."""
        import re

        for pat, conv in ((r"(.*\.pre)\.norm_layers\.(\d+)\.", r"\1.conv_layers.\2.bias"),
                          (r"(.*)\.norm_layers_1\.(\d+)\.", r"\1.attn_layers.\2.conv_o.bias"),
                          (r"(.*)\.norm_layers_2\.(\d+)\.", r"\1.ffn_layers.\2.conv_2.bias"),
                          (r"(.*\.proj_w)\.norm_(\d+)\.", r"\1.conv_\2.bias")):
            m = re.match(pat, norm_name)
            if m:
                b = named.get(m.expand(conv))
                return int(np.asarray(b).size) if b is not None else None
        m = re.match(r"(decoder\.flows)\.(\d+)\.(logs|bias)$", norm_name)
        if m:  # ActNorm of block b sits two flows before that block's coupling layer, whose `end` conv has as many rows
            b = named.get(f"{m.group(1)}.{int(m.group(2)) + 2}.end.bias")
            return int(np.asarray(b).size) if b is not None else None
        return None

    recovered: typing.List[str] = []
    need_inv = [m for m in missing if m.endswith(".weight_inv")]
    need_gamma = [m for m in missing if m.endswith(".gamma")]
    need_beta = [m for m in missing if m.endswith(".beta")]
    need_logs = [m for m in missing if m.endswith(".logs")]
    need_anb = [m for m in missing if m.endswith(".bias") and m[: -len(".bias")] + ".logs" in need_logs]
    other = [m for m in missing if m not in need_inv + need_gamma + need_beta + need_logs + need_anb]
    if other:
        # a needed tensor the reader saw but could not take (external data, element-count mismatch): say THAT
        why = [msg for raw, msg in G.g.unreadable.items() if any(_strip_prefix(raw) in (o, o[:-7] + ".weight_g", o[:-7] + ".weight_v") for o in other)]
        if why:
            raise ValueError(f"{path}: {why[0]}")
        raise KeyError(f"{path}: no tensor for {other[:4]}{' ...' if len(other) > 4 else ''} (exporter layout not recognised)")
    if need_inv:
        # the manifest lists blocks 0..n-1; the reverse pass executes n-1..0
        blocks = sorted(need_inv, key=lambda s: int(s.split(".")[2]))
        if len(inv_weights) != len(blocks):
            raise KeyError(f"{path}: {len(inv_weights)} {n_split}x{n_split} convs in the graph, {len(blocks)} InvConvNear weights needed")
        for name, w in zip(reversed(blocks), inv_weights):
            out[name] = w.reshape(n_split, n_split)
            recovered.append(name)
    if need_gamma or need_beta:
        if len(need_gamma) != len(need_beta) or len(ln_pairs) < len(need_gamma):
            raise KeyError(f"{path}: {len(ln_pairs)} LayerNorm scale/shift pairs in the graph, {len(need_gamma)} needed")
        all_ln = [m for m in manifest_names if m.endswith(".gamma")]
        if len(ln_pairs) != len(all_ln):
            raise KeyError(f"{path}: {len(ln_pairs)} LayerNorm patterns, model has {len(all_ln)} LayerNorms")
        for k, gname in enumerate(all_ln):
            want = anchor_size(gname)
            if want is not None and ln_pairs[k][0].size != want:
                raise KeyError(f"{path}: LayerNorm pattern #{k} has {ln_pairs[k][0].size} channels but '{gname}' follows a conv of "
                               f"{want}: the graph's Mul/Add order does not match the model (exporter layout not recognised)")
            if gname in need_gamma:
                out[gname] = ln_pairs[k][0].reshape(-1)
                out[gname[: -len("gamma")] + "beta"] = ln_pairs[k][1].reshape(-1)
                recovered += [gname, gname[: -len("gamma")] + "beta"]
    if need_logs:
        all_an = sorted([m for m in manifest_names if m.endswith(".logs")], key=lambda s: int(s.split(".")[2]))
        if len(an_pairs) != len(all_an):
            raise KeyError(f"{path}: {len(an_pairs)} ActNorm patterns in the graph, model has {len(all_an)}")
        for lname, (bias, logs) in zip(reversed(all_an), an_pairs):
            want = anchor_size(lname)
            if want is not None and logs.size != want:
                raise KeyError(f"{path}: ActNorm pattern for '{lname}' has {logs.size} channels, its coupling block has {want} "
                               "(exporter layout not recognised)")
            if lname in need_logs:
                recovered.append(lname)
                out[lname] = logs.reshape(1, -1, 1)
                bname = lname[: -len("logs")] + "bias"
                if bname not in out:
                    out[bname] = bias.reshape(1, -1, 1)
                    recovered.append(bname)
    if recovered:
        import logging

        logging.getLogger("larynx_amd.onnx").info("%s: %d tensors had lost their names to constant folding and were recovered by graph "
                                                  "pattern, in execution order: %s", path, len(recovered), ", ".join(recovered))
    return out
