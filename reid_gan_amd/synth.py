"""Deterministic synthetic weights and inputs.

There is no network for checkpoints or datasets, so parity tests and the
benchmark regenerate every state dict from a seed.  numpy's PCG64 stream is
used (not torch's RNG) so that the build container and the GPU box produce
identical values.  Pure numpy/torch-CPU; no GPU code here.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def kind_of(name: str, shape: Tuple[int, ...]) -> str:
    if name.endswith("num_batches_tracked"):
        return "counter"
    if name.endswith("running_mean"):
        return "running_mean"
    if name.endswith("running_var"):
        return "running_var"
    if name.endswith("weight_u") or name.endswith("weight_v"):
        return "sn_vec"
    if name.endswith("gamma"):
        return "gamma"
    if len(shape) == 4:
        return "conv"
    if name.endswith("bias"):
        return "bn_bias" if (".bn." in name or "identity_downsample.1" in name) else "conv_bias"
    return "bn_weight"


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    kind = kind_of(name, shape)
    r = _rng(seed, name)
    if kind == "counter":
        return torch.zeros(shape, dtype=torch.long)
    if kind == "conv":
        fan_in = shape[1] * shape[2] * shape[3]
        if ".deconv." in name:
            fan_in = shape[0] * shape[2] * shape[3] / 4.0
        a = r.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        if "query_conv" in name or "key_conv" in name:
            a = a * 0.004         # the discriminator sees raw HU (|x| ~ 1e2): keep q.k energies O(1) so the
                                  # softmax is not a hard argmax, which no finite-precision parity survives
    elif kind == "running_mean":
        a = 0.1 * r.standard_normal(shape)
    elif kind == "running_var":
        a = 1.0 + 0.2 * r.random(shape)
    elif kind == "sn_vec":
        a = r.standard_normal(shape)
        a = a / max(np.linalg.norm(a), 1e-12)
    elif kind == "gamma":
        a = np.full(shape, 0.7)
    elif kind == "bn_bias":
        a = 0.2 * r.standard_normal(shape)
    elif kind == "conv_bias":
        a = 0.05 * r.standard_normal(shape)
    else:  # bn_weight
        a = 1.0 + 0.2 * r.standard_normal(shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_state(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, torch.Tensor]:
    """spec: iterable of (name, shape[, ...])."""
    return {s[0]: synth_tensor(s[0], tuple(s[1]), seed) for s in spec}


VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]


def vgg_spec():
    out, idx, cin = [], 0, 3
    for v in VGG_CFG:
        if v == "M":
            idx += 1
        else:
            out.append((f"{idx}.weight", (v, cin, 3, 3)))
            out.append((f"{idx}.bias", (v,)))
            cin = v
            idx += 2
    return out


def synth_vgg(seed: int = 16) -> Dict[str, torch.Tensor]:
    """Seeded stand-in for the ImageNet VGG16 feature weights (keys '<features idx>.weight|bias')."""
    return synth_state(vgg_spec(), seed)


def synth_batch(batch: int, size: int, seed: int = 1234):
    """HU-like target y and noisier input x, fp32 NCHW (B,1,size,size):
    y = clamp(-470 + 520 z1, -1024, 3072), x = clamp(y + 30 z2, -1024, 3072)."""
    r = np.random.default_rng(int(seed))
    y = np.clip(-470.0 + 520.0 * r.standard_normal((batch, 1, size, size)), -1024.0, 3072.0)
    x = np.clip(y + 30.0 * r.standard_normal((batch, 1, size, size)), -1024.0, 3072.0)
    return torch.from_numpy(x.astype(np.float32)), torch.from_numpy(y.astype(np.float32))
