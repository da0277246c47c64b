"""Drop-in metric API of the reference's src/measure.py on the MI355X kernels.

compute_measure(x, y, pred, data_range) returns the same two (PSNR, SSIM, RMSE) tuples of Python floats.  The
reference launches ten convolutions and ~40 elementwise ops over six host synchronisations; here one fused kernel
reads x, y and pred once (12 bytes per pixel) and one 16-byte copy brings the four sums back.
Inputs may be torch tensors on any device or numpy arrays (as the reference accepts); they are moved to the GPU.
compute_measure_batch is the batched form (one launch for N slices) used by the inference engine.
"""
from __future__ import annotations

import math
from math import exp

import numpy as np
import torch

from . import ops

F32 = torch.float32


def gaussian(window_size, sigma):
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _img(t) -> torch.Tensor:
    """fp32 contiguous (N, H, W) device tensor from a 2-d / 3-d / 4-d (one channel) tensor or ndarray."""
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    if t.dim() == 2:
        t = t.reshape(1, *t.shape)
    elif t.dim() == 4:
        if t.shape[1] != 1:
            raise NotImplementedError("metric kernels take one-channel images")
        t = t.reshape(t.shape[0], t.shape[2], t.shape[3])
    return t.detach().to(device="cuda", dtype=F32).contiguous()


def _pair_sums(a, b, data_range, three=None):
    """Per-image [sum ssim(a,b), sum (a-b)^2, (sum ssim(three,b), sum (three-b)^2)] and the pixel count."""
    ops.require_gpu()
    a, b = _img(a), _img(b)
    t = None if three is None else _img(three)
    sums = torch.zeros(a.shape[0], 4, dtype=F32, device="cuda")
    ops.ssim_fwd(a, t, b, (0.01 * data_range) ** 2, (0.03 * data_range) ** 2, sums)
    return sums.double().cpu(), a.shape[1] * a.shape[2]


def compute_measure_batch(x, y, pred, data_range):
    """[( (psnr, ssim, rmse) of x vs y, (psnr, ssim, rmse) of pred vs y )] per image, one kernel launch."""
    s, npx = _pair_sums(x, y, data_range, three=pred)
    out = []
    for i in range(s.shape[0]):
        mo, mp = s[i, 1].item() / npx, s[i, 3].item() / npx
        out.append(((10 * math.log10(data_range ** 2 / mo), s[i, 0].item() / npx, math.sqrt(mo)),
                    (10 * math.log10(data_range ** 2 / mp), s[i, 2].item() / npx, math.sqrt(mp))))
    return out


def compute_measure(x, y, pred, data_range):
    """reference: src/measure.py:8-15."""
    res = compute_measure_batch(x, y, pred, data_range)
    if len(res) != 1:
        raise ValueError("compute_measure takes one image triple; use compute_measure_batch for a batch")
    return res[0]


def compute_MSE(img1, img2):
    """mean((img1 - img2)^2) as a Python float (reference: src/measure.py:18-19)."""
    s, npx = _pair_sums(img1, img2, 1.0)
    return s[:, 1].sum().item() / (npx * s.shape[0])


def compute_RMSE(img1, img2):
    return math.sqrt(compute_MSE(img1, img2))                       # src/measure.py:22-26


def compute_PSNR(img1, img2, data_range):
    mse = compute_MSE(img1, img2)
    if mse == 0:                         # identical images: log10 of a division by zero (src/measure.py:31-32)
        return float("inf")
    return 10 * math.log10((data_range ** 2) / mse)     # src/measure.py:29-35


def compute_SSIM(img1, img2, data_range, window_size=11, channel=1, size_average=True):
    """reference: src/measure.py:49-74 (C1 = (0.01*range)^2, C2 = (0.03*range)^2)."""
    if window_size != 11 or channel != 1:
        raise NotImplementedError("the fused SSIM kernel implements window_size=11, channel=1")
    s, npx = _pair_sums(img1, img2, data_range)
    per = s[:, 0] / npx
    if size_average:
        return per.mean().item()
    if per.numel() != 1:
        raise ValueError("size_average=False returns one value only for a single image (as .item() in the reference)")
    return per[0].item()


def compute_CNR(img1, img2):
    """(mean(img1) - mean(img2)) / std(img2), population std.  The reference's tensor branch raises (it passes
    `unbiased` to torch.mean, src/measure.py:39); this follows its numpy branch (src/measure.py:44-47)."""
    ops.require_gpu()
    a, b = _img(img1), _img(img2)
    sc = torch.zeros(3, dtype=F32, device="cuda")
    ops.sum_f32(a, a.numel(), 1, 1.0 / a.numel(), sc[0:])
    ops.sum_f32(b, b.numel(), 1, 1.0 / b.numel(), sc[1:])
    ops.sum_f32(b, b.numel(), 1, 1.0 / b.numel(), sc[2:], square=True)
    m1, m2, e2 = sc.double().cpu().tolist()
    return (m1 - m2) / math.sqrt(max(e2 - m2 * m2, 0.0))
