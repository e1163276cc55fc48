"""TEST INFRASTRUCTURE — the HiFi-GAN oracle restated on torch CPU operators.

Same algorithm as `oracle/hifi_gan_np.py` (a restatement of
`hifi_gan/models.py:186-202`, NOT the reference's module classes — /root/reference
does not exist on the GPU box), but with `torch.nn.functional.conv1d /
conv_transpose1d` doing the arithmetic, i.e. the same oneDNN CPU kernels the
reference's own `--backend pytorch` path runs on.  Used only as the timed
`cpu_baseline` of bench.py (the numpy oracle is 10-50x slower than any real CPU
deployment and would flatter the GPU); pinned against the numpy oracle in
tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np

from . import hifi_gan_np as ref_np


def hifigan_infer_torch(sd, hp, mel: np.ndarray, threads: int = 0) -> np.ndarray:
    import torch
    import torch.nn.functional as F

    if threads:
        torch.set_num_threads(threads)

    def w(prefix):
        return torch.from_numpy(np.ascontiguousarray(ref_np._w(sd, prefix)))

    def b(prefix):
        return torch.from_numpy(np.ascontiguousarray(ref_np._b(sd, prefix)))

    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(mel, np.float32))[None]
        x = F.conv1d(x, w("conv_pre"), b("conv_pre"), padding=3)
        nk = len(hp.resblock_kernel_sizes)
        for i, (u, ku) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
            x = F.leaky_relu(x, 0.1)
            x = F.conv_transpose1d(x, w(f"ups.{i}"), b(f"ups.{i}"), stride=u, padding=(ku - u) // 2)
            xs = None
            for j, (k, dil) in enumerate(zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes)):
                p = f"resblocks.{i * nk + j}"
                r = x
                for m, d in enumerate(dil):
                    if hp.resblock == "1":
                        xt = F.conv1d(F.leaky_relu(r, 0.1), w(f"{p}.convs1.{m}"), b(f"{p}.convs1.{m}"), dilation=d, padding=ref_np.get_padding(k, d))
                        xt = F.conv1d(F.leaky_relu(xt, 0.1), w(f"{p}.convs2.{m}"), b(f"{p}.convs2.{m}"), padding=ref_np.get_padding(k, 1))
                    else:
                        xt = F.conv1d(F.leaky_relu(r, 0.1), w(f"{p}.convs.{m}"), b(f"{p}.convs.{m}"), dilation=d, padding=ref_np.get_padding(k, d))
                    r = xt + r
                xs = r if xs is None else xs + r
            x = xs / nk
        x = F.leaky_relu(x)
        x = torch.tanh(F.conv1d(x, w("conv_post"), b("conv_post"), padding=3))
        return x[0, 0].numpy()
