"""CPU oracle for the RIED-GAN hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-tensor restatement (stock torch CPU ops, fp32) of the
arithmetic of the reference's generator, discriminator, losses, metrics and
training step.  It exists to check the HIP path; nothing in the product
package imports it.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may use it.

Parity pinning
  * generator / discriminator / blocks / metrics: pinned by golden vectors
    made by importing the reference's src/rinet.py and src/measure.py
    (tests/golden/make_golden.py, run in the build container).
  * src/loss.py cannot be imported here (it imports torchvision at module
    scope, and torchvision is not installed).  Its compute_SSIM is the same
    arithmetic as measure.compute_SSIM with C1=0.01**2, C2=0.03**2, i.e.
    data_range == 1, and is pinned through that.  DLoss / GLoss /
    CompoundLoss weighting are restated from the text of loss.py.
  * VGG16 perceptual taps: the layer table is the published VGG16
    configuration D; pretrained ImageNet weights are not available offline,
    so parity for the perceptual term is "unpinned" (checked against this
    restatement with seeded synthetic weights only).

All functions take parameters as a flat dict keyed by the reference's
state_dict names, so a reference checkpoint's netG_state_dict /
netD_state_dict can be passed directly.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ---------------------------------------------------------------------------
# optional emulation of the product's storage precision: with EMULATE_BF16 set, every tensor the HIP path keeps
# in bf16 (packed weights, pre-BN conv outputs, activations, attention probabilities) is rounded to bf16 at the
# same point, while all arithmetic stays fp32 as in the MFMA accumulators.  Off by default: the plain oracle is
# the fp32 reference arithmetic.  Tests use both: tight tolerance against the emulation, loose against fp32.
# ---------------------------------------------------------------------------
EMULATE_BF16 = False


def set_bf16_emulation(flag: bool) -> None:
    global EMULATE_BF16
    EMULATE_BF16 = bool(flag)


def _r(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32) if EMULATE_BF16 else t


BN_EPS = 1e-5          # nn.BatchNorm2d default; eps=0.001 is commented out (rinet.py:187,199)
BN_MOMENTUM = 0.1
SN_EPS = 1e-12         # torch.nn.utils.spectral_norm default


# --------------------------------------------------------------------------
# parameter access with declared shapes (also used to enumerate the spec)
# --------------------------------------------------------------------------
class Params:
    """Dict wrapper.  get(name, shape) checks the shape; in spec mode it
    records (name, shape, kind) and hands back zeros so a dry forward
    enumerates every key the reference's state_dict holds."""

    def __init__(self, sd: Optional[Dict[str, Tensor]] = None, spec_mode: bool = False):
        self.sd = {} if sd is None else sd
        self.spec_mode = spec_mode
        self.spec: List[Tuple[str, Tuple[int, ...], str]] = []

    def get(self, name: str, shape, kind: str = "param") -> Tensor:
        shape = tuple(int(s) for s in shape)
        if self.spec_mode:
            if name not in self.sd:
                self.spec.append((name, shape, kind))
                if kind == "running_var":
                    self.sd[name] = torch.ones(shape)
                elif kind == "counter":
                    self.sd[name] = torch.zeros(shape, dtype=torch.long)
                else:
                    self.sd[name] = torch.full(shape, 0.01)
            return self.sd[name]
        t = self.sd[name]
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
        return t

    def set(self, name: str, value: Tensor) -> None:
        self.sd[name] = value


# --------------------------------------------------------------------------
# generator (src/rinet.py:34-129, 183-445)
# --------------------------------------------------------------------------
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _bn_relu(P: Params, pre: str, x: Tensor, c: int, training: bool, relu: bool = True,
             round_out: bool = True) -> Tensor:
    """BatchNorm2d (+ReLU).  Train mode uses batch statistics and updates the
    running stats with momentum 0.1 and the unbiased variance (torch semantics;
    rinet.py:187-193)."""
    w = P.get(pre + ".weight", (c,))
    b = P.get(pre + ".bias", (c,))
    rm = P.get(pre + ".running_mean", (c,), "running_mean")
    rv = P.get(pre + ".running_var", (c,), "running_var")
    nbt = P.get(pre + ".num_batches_tracked", (), "counter")
    if training:
        xd = x.detach()
        n = xd.numel() // c
        mean = xd.mean(dim=(0, 2, 3))
        var_b = xd.var(dim=(0, 2, 3), unbiased=False)
        var_u = var_b * (n / max(n - 1, 1))
        P.set(pre + ".running_mean", (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean)
        P.set(pre + ".running_var", (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_u)
        P.set(pre + ".num_batches_tracked", nbt + 1)
        if EMULATE_BF16:   # statistics from the fp32 accumulators, normalisation of the bf16-stored conv output
            m_g = x.mean(dim=(0, 2, 3))                                  # attached: gradients flow through the stats
            istd = torch.rsqrt(x.var(dim=(0, 2, 3), unbiased=False) + BN_EPS)
            y = (_r(x) - m_g.view(1, -1, 1, 1)) * (w * istd).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        else:
            y = F.batch_norm(x, None, None, w, b, True, BN_MOMENTUM, BN_EPS)
    else:
        y = F.batch_norm(_r(x), rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)
    y = F.relu(y) if relu else y
    return _r(y) if round_out else y      # a branch that is added onto the residual is rounded after the add


def basic_conv(P: Params, pre: str, x: Tensor, cin: int, cout: int, k, stride=1, pad=0,
               training: bool = True, final: bool = False) -> Tensor:
    """BasicConv2d: conv(no bias) -> BN -> ReLU (rinet.py:183-193)."""
    kh, kw = _pair(k)
    w = P.get(pre + ".conv.weight", (cout, cin, kh, kw))
    if cin > 1:          # the one-channel first layer runs in fp32 on the raw image
        w = _r(w)
    y = F.conv2d(x, w, None, stride, _pair(pad))
    return _bn_relu(P, pre + ".bn", y, cout, training, round_out=not final)


def basic_deconv(P: Params, pre: str, x: Tensor, cin: int, cout: int, k, stride=1, pad=0,
                 training: bool = True) -> Tensor:
    """BasicDeconv2d: ConvTranspose2d(no bias) -> BN -> ReLU (rinet.py:195-205).
    The weight is laid out (Cin, Cout, kh, kw)."""
    kh, kw = _pair(k)
    w = _r(P.get(pre + ".deconv.weight", (cin, cout, kh, kw)))
    y = F.conv_transpose2d(x, w, None, stride, _pair(pad))
    return _bn_relu(P, pre + ".bn", y, cout, training)


def _identity_path(P: Params, pre: str, x: Tensor, cin: int, cout: int, has_ds: bool, training: bool) -> Tensor:
    """identity_downsample = Conv1x1(no bias) + BN + ReLU (rinet.py:98-103)."""
    if not has_ds:
        return x
    w = _r(P.get(pre + ".identity_downsample.0.weight", (cout, cin, 1, 1)))
    y = F.conv2d(x, w)
    return _bn_relu(P, pre + ".identity_downsample.1", y, cout, training)


def inception_a(P: Params, pre: str, x: Tensor, cin: int, has_ds: bool, training: bool = True) -> Tensor:
    """rinet.py:207-252 -> 32|32|48|16 = 128 channels, plus identity, no activation after the sum."""
    t = training
    b1 = basic_conv(P, pre + ".branch1x1", x, cin, 32, 1, training=t, final=True)
    b3 = basic_conv(P, pre + ".branch3x3_1", x, cin, 16, 1, training=t)
    b3 = basic_conv(P, pre + ".branch3x3_2", b3, 16, 32, 3, pad=1, training=t, final=True)
    bd = basic_conv(P, pre + ".branch3x3dbl_1", x, cin, 32, 1, training=t)
    bd = basic_conv(P, pre + ".branch3x3dbl_2", bd, 32, 48, 3, pad=1, training=t)
    bd = basic_conv(P, pre + ".branch3x3dbl_3", bd, 48, 48, 3, pad=1, training=t, final=True)
    bp = _r(F.avg_pool2d(x, 3, 1, 1))
    bp = basic_conv(P, pre + ".branch_pool", bp, cin, 16, 1, training=t, final=True)
    out = torch.cat([b1, b3, bd, bp], 1)
    return _r(out + _identity_path(P, pre, x, cin, 128, has_ds, t))


def inception_b(P: Params, pre: str, x: Tensor, cin: int, decoder: bool, training: bool = True) -> Tensor:
    """rinet.py:254-291.  Encoder: 3x3 s2 convs + maxpool(3,2,1).
    Decoder: BasicDeconv2d everywhere (k=4 s2 for the strided ones) and
    sample = conv1x1(cin->128, no bias) + bilinear x2 (align_corners=True)."""
    t = training
    blk = basic_deconv if decoder else basic_conv
    k = 4 if decoder else 3
    b3 = blk(P, pre + ".branch3x3_1", x, cin, 32, 1, training=t)
    b3 = blk(P, pre + ".branch3x3_2", b3, 32, 64, k, stride=2, pad=1, training=t)
    bd = blk(P, pre + ".branch3x3dbl_1", x, cin, 32, 1, training=t)
    bd = blk(P, pre + ".branch3x3dbl_2", bd, 32, 48, 3, pad=1, training=t)
    bd = blk(P, pre + ".branch3x3dbl_3", bd, 48, 64, k, stride=2, pad=1, training=t)
    if decoder:
        w = _r(P.get(pre + ".sample.0.weight", (128, cin, 1, 1)))
        bp = _r(F.interpolate(_r(F.conv2d(x, w)), scale_factor=2, mode="bilinear", align_corners=True))
    else:
        bp = F.max_pool2d(x, 3, 2, 1)
    return torch.cat([b3, bd, bp], 1)


def inception_c(P: Params, pre: str, x: Tensor, cin: int, c7: int, has_ds: bool, training: bool = True) -> Tensor:
    """rinet.py:293-347 -> 4 x 64 = 256 channels plus identity."""
    t = training
    b1 = basic_conv(P, pre + ".branch1x1", x, cin, 64, 1, training=t, final=True)
    b7 = basic_conv(P, pre + ".branch7x7_1", x, cin, c7, 1, training=t)
    b7 = basic_conv(P, pre + ".branch7x7_2", b7, c7, c7, (1, 7), pad=(0, 3), training=t)
    b7 = basic_conv(P, pre + ".branch7x7_3", b7, c7, 64, (7, 1), pad=(3, 0), training=t, final=True)
    bd = basic_conv(P, pre + ".branch7x7dbl_1", x, cin, c7, 1, training=t)
    bd = basic_conv(P, pre + ".branch7x7dbl_2", bd, c7, c7, (7, 1), pad=(3, 0), training=t)
    bd = basic_conv(P, pre + ".branch7x7dbl_3", bd, c7, c7, (1, 7), pad=(0, 3), training=t)
    bd = basic_conv(P, pre + ".branch7x7dbl_4", bd, c7, c7, (7, 1), pad=(3, 0), training=t)
    bd = basic_conv(P, pre + ".branch7x7dbl_5", bd, c7, 64, (1, 7), pad=(0, 3), training=t, final=True)
    bp = _r(F.avg_pool2d(x, 3, 1, 1))
    bp = basic_conv(P, pre + ".branch_pool", bp, cin, 64, 1, training=t, final=True)
    out = torch.cat([b1, b7, bd, bp], 1)
    return _r(out + _identity_path(P, pre, x, cin, 256, has_ds, t))


def inception_d(P: Params, pre: str, x: Tensor, cin: int, decoder: bool, training: bool = True) -> Tensor:
    """rinet.py:349-387."""
    t = training
    blk = basic_deconv if decoder else basic_conv
    k = 4 if decoder else 3
    b3 = blk(P, pre + ".branch3x3_1", x, cin, 64, 1, training=t)
    b3 = blk(P, pre + ".branch3x3_2", b3, 64, 128, k, stride=2, pad=1, training=t)
    b7 = blk(P, pre + ".branch7x7x3_1", x, cin, 128, 1, training=t)
    b7 = blk(P, pre + ".branch7x7x3_2", b7, 128, 128, (1, 7), pad=(0, 3), training=t)
    b7 = blk(P, pre + ".branch7x7x3_3", b7, 128, 128, (7, 1), pad=(3, 0), training=t)
    b7 = blk(P, pre + ".branch7x7x3_4", b7, 128, 128, k, stride=2, pad=1, training=t)
    if decoder:
        w = _r(P.get(pre + ".sample.0.weight", (256, cin, 1, 1)))
        bp = _r(F.interpolate(_r(F.conv2d(x, w)), scale_factor=2, mode="bilinear", align_corners=True))
    else:
        bp = F.max_pool2d(x, 3, 2, 1)
    return torch.cat([b3, b7, bp], 1)


def inception_e(P: Params, pre: str, x: Tensor, cin: int, has_ds: bool, training: bool = True) -> Tensor:
    """rinet.py:389-445 -> 160 | 192+192 | 192+192 | 96 = 1024 channels plus identity."""
    t = training
    b1 = basic_conv(P, pre + ".branch1x1", x, cin, 160, 1, training=t, final=True)
    b3 = basic_conv(P, pre + ".branch3x3_1", x, cin, 192, 1, training=t)
    b3a = basic_conv(P, pre + ".branch3x3_2a", b3, 192, 192, (1, 3), pad=(0, 1), training=t, final=True)
    b3b = basic_conv(P, pre + ".branch3x3_2b", b3, 192, 192, (3, 1), pad=(1, 0), training=t, final=True)
    bd = basic_conv(P, pre + ".branch3x3dbl_1", x, cin, 224, 1, training=t)
    bd = basic_conv(P, pre + ".branch3x3dbl_2", bd, 224, 192, 3, pad=1, training=t)
    bda = basic_conv(P, pre + ".branch3x3dbl_3a", bd, 192, 192, (1, 3), pad=(0, 1), training=t, final=True)
    bdb = basic_conv(P, pre + ".branch3x3dbl_3b", bd, 192, 192, (3, 1), pad=(1, 0), training=t, final=True)
    bp = _r(F.avg_pool2d(x, 3, 1, 1))
    bp = basic_conv(P, pre + ".branch_pool", bp, cin, 96, 1, training=t, final=True)
    out = torch.cat([b1, b3a, b3b, bda, bdb, bp], 1)
    return _r(out + _identity_path(P, pre, x, cin, 1024, has_ds, t))


G_TOP_ACTS = ["conv1", "conv2", "layer_block_1", "reduce_grid_1", "layer_block_2", "reduce_grid_2",
              "layer_block_3", "expand_grid_1", "layer_block_4", "expand_grid_2", "layer_block_5",
              "deconv1", "deconv2", "out"]


def rigan_forward(P: Params, x: Tensor, training: bool = True,
                  acts: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """RIGAN.forward (rinet.py:72-92).  test.py never calls eval(), so the
    reference runs this in train mode at inference too (test.py:112-131)."""
    t = training
    a = {} if acts is None else acts
    a["conv1"] = c1 = basic_conv(P, "conv1", x, 1, 32, 3, 2, 1, t)
    a["conv2"] = c2 = basic_conv(P, "conv2", c1, 32, 64, 3, 2, 1, t)
    a["layer_block_1"] = h = inception_a(P, "layer_block_1.0", c2, 64, True, t)
    a["reduce_grid_1"] = h = inception_b(P, "reduce_grid_1", h, 128, False, t)
    a["layer_block_2"] = h = inception_c(P, "layer_block_2.0", h, 256, 64, False, t)
    a["reduce_grid_2"] = h = inception_d(P, "reduce_grid_2", h, 256, False, t)
    a["layer_block_3"] = h = inception_e(P, "layer_block_3.0", h, 512, True, t)
    a["expand_grid_1"] = h = inception_d(P, "expand_grid_1", h, 1024, True, t)
    a["layer_block_4"] = h = inception_c(P, "layer_block_4.0", h, 512, 64, True, t)
    a["expand_grid_2"] = h = inception_b(P, "expand_grid_2", h, 256, True, t)
    a["layer_block_5"] = h = inception_a(P, "layer_block_5.0", h, 256, True, t)
    a["deconv1"] = h = basic_deconv(P, "deconv1", torch.cat([c2, h], 1), 192, 64, 4, 2, 1, t)
    a["deconv2"] = h = basic_deconv(P, "deconv2", torch.cat([c1, h], 1), 96, 32, 4, 2, 1, t)
    w = P.get("out_conv.weight", (1, 33, 1, 1))
    a["out"] = out = F.conv2d(torch.cat([x, h], 1), w)
    return out


def rigan_spec() -> List[Tuple[str, Tuple[int, ...], str]]:
    P = Params(spec_mode=True)
    with torch.no_grad():
        rigan_forward(P, torch.zeros(2, 1, 32, 32), training=True)
    return P.spec


# --------------------------------------------------------------------------
# discriminator (src/rinet.py:10-32, 131-181)
# --------------------------------------------------------------------------
def sn_weight(P: Params, pre: str, cout: int, cin: int, k: int, training: bool) -> Tensor:
    """Spectral-normalised weight with one power iteration in train mode
    (torch/nn/utils/spectral_norm.py:92-114; applied at rinet.py:138).
    u and v are updated in place (here: written back to the dict) and are
    constants for autograd; sigma = u . (W v) carries gradient to weight_orig."""
    w = P.get(pre + ".weight_orig", (cout, cin, k, k))
    u = P.get(pre + ".weight_u", (cout,), "sn_u")
    v = P.get(pre + ".weight_v", (cin * k * k,), "sn_v")
    wm = w.reshape(cout, -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=SN_EPS)
        P.set(pre + ".weight_u", u)
        P.set(pre + ".weight_v", v)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def sn_conv(P: Params, pre: str, x: Tensor, cin: int, cout: int, stride: int, training: bool) -> Tensor:
    """SNConv2d: SN conv 3x3 pad 1 with bias, LeakyReLU(0.2) (rinet.py:131-148)."""
    w = sn_weight(P, pre + ".conv2d", cout, cin, 3, training)
    if cin > 1:
        w = _r(w)
    b = P.get(pre + ".conv2d.bias", (cout,))
    return _r(F.leaky_relu(F.conv2d(x, w, b, stride, 1), 0.2))


def self_attn(P: Params, pre: str, x: Tensor, c: int) -> Tensor:
    """Self_Attn (rinet.py:150-181): softmax over keys of q^T k; out = gamma * (v A^T) + x."""
    B, C, H, W = x.shape
    n = H * W
    wq = P.get(pre + ".query_conv.weight", (c // 8, c, 1, 1)); bq = P.get(pre + ".query_conv.bias", (c // 8,))
    wk = P.get(pre + ".key_conv.weight", (c // 8, c, 1, 1)); bk = P.get(pre + ".key_conv.bias", (c // 8,))
    wv = P.get(pre + ".value_conv.weight", (c, c, 1, 1)); bv = P.get(pre + ".value_conv.bias", (c,))
    gamma = P.get(pre + ".gamma", (1,))
    q = _r(F.conv2d(x, _r(wq), bq)).view(B, -1, n).permute(0, 2, 1)      # B x N x d
    k = _r(F.conv2d(x, _r(wk), bk)).view(B, -1, n)                       # B x d x N
    att = _r(torch.softmax(torch.bmm(q, k), dim=-1))                     # B x N x N
    v = _r(F.conv2d(x, _r(wv), bv)).view(B, -1, n)                       # B x C x N
    out = _r(torch.bmm(v, att.permute(0, 2, 1))).view(B, C, H, W)
    return _r(gamma * out + x)


def discriminator_forward(P: Params, x: Tensor, training: bool = True,
                          acts: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """ImageDiscriminator.forward (rinet.py:19-32): returns (B, 512*(H/16)*(W/16))."""
    a = {} if acts is None else acts
    p = "discriminator_net."
    a["d0"] = h = sn_conv(P, p + "0", x, 1, 64, 1, training)
    a["d1"] = h = sn_conv(P, p + "1", h, 64, 128, 2, training)
    a["d2"] = h = sn_conv(P, p + "2", h, 128, 256, 2, training)
    a["d3"] = h = self_attn(P, p + "3", h, 256)
    a["d4"] = h = sn_conv(P, p + "4", h, 256, 512, 2, training)
    a["d5"] = h = self_attn(P, p + "5", h, 512)
    a["d6"] = h = sn_conv(P, p + "6", h, 512, 512, 2, training)
    return h.reshape(h.size(0), -1)


def discriminator_spec() -> List[Tuple[str, Tuple[int, ...], str]]:
    P = Params(spec_mode=True)
    with torch.no_grad():
        discriminator_forward(P, torch.zeros(1, 1, 16, 16), training=False)
    return P.spec


# --------------------------------------------------------------------------
# SSIM / metrics (src/loss.py:24-67, src/measure.py:8-86)
# --------------------------------------------------------------------------
def gaussian_1d(window_size: int = 11, sigma: float = 1.5) -> Tensor:
    """loss.py:24-26 / measure.py:77-79 (built in Python double, stored fp32, normalised in fp32)."""
    g = torch.tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)],
                     dtype=torch.float32)
    return g / g.sum()


def gaussian_window(window_size: int = 11) -> Tensor:
    g = gaussian_1d(window_size).unsqueeze(1)
    return g.mm(g.t()).float().reshape(1, 1, window_size, window_size)


def ssim_map(img1: Tensor, img2: Tensor, c1: float, c2: float, window_size: int = 11) -> Tensor:
    """The five zero-padded Gaussian filters and the SSIM quotient, with the
    sigma = E[x^2] - mu^2 form of the reference (loss.py:51-62, measure.py:58-70)."""
    if img1.dim() == 2:
        img1 = img1.reshape(1, 1, *img1.shape)
        img2 = img2.reshape(1, 1, *img2.shape)
    w = gaussian_window(window_size).to(img1.dtype)
    p = window_size // 2
    mu1 = F.conv2d(img1, w, padding=p)
    mu2 = F.conv2d(img2, w, padding=p)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=p) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=p) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=p) - mu12
    return ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))


def loss_ssim(img1: Tensor, img2: Tensor) -> Tensor:
    """loss.compute_SSIM with size_average=True: C1=0.01**2, C2=0.03**2 on RAW values (loss.py:60-65)."""
    return ssim_map(img1, img2, 0.01 ** 2, 0.03 ** 2).mean()


def dssim_loss(target: Tensor, pred: Tensor) -> Tensor:
    """loss.SSIM.forward(target, pred) = (1 - ssim)/2 (loss.py:125-130)."""
    return (1.0 - loss_ssim(target, pred)) / 2.0


def measure_ssim(img1: Tensor, img2: Tensor, data_range: float) -> float:
    """measure.compute_SSIM (measure.py:49-74)."""
    return ssim_map(img1, img2, (0.01 * data_range) ** 2, (0.03 * data_range) ** 2).mean().item()


def measure_mse(a: Tensor, b: Tensor) -> Tensor:
    return ((a - b) ** 2).mean()                                  # measure.py:18-19


def measure_rmse(a: Tensor, b: Tensor) -> float:
    return torch.sqrt(measure_mse(a, b)).item()                   # measure.py:22-26


def measure_psnr(a: Tensor, b: Tensor, data_range: float) -> float:
    return 10 * torch.log10((data_range ** 2) / measure_mse(a, b)).item()   # measure.py:29-35


def measure_cnr(a: Tensor, b: Tensor) -> float:
    """measure.compute_CNR.  The reference's tensor branch raises (it passes
    unbiased= to torch.mean, measure.py:39); this follows its numpy branch
    (measure.py:44-47): population std."""
    return ((a.mean() - b.mean()) / b.std(unbiased=False)).item()


def compute_measure(x: Tensor, y: Tensor, pred: Tensor, data_range: float):
    """measure.compute_measure (measure.py:8-15)."""
    return ((measure_psnr(x, y, data_range), measure_ssim(x, y, data_range), measure_rmse(x, y)),
            (measure_psnr(pred, y, data_range), measure_ssim(pred, y, data_range), measure_rmse(pred, y)))


def trunc(t: Tensor, lo: float = -160.0, hi: float = 240.0) -> Tensor:
    return t.clamp(lo, hi)                                        # test.py:77-80


# --------------------------------------------------------------------------
# losses (src/loss.py:20-22, 69-87, 132-197)
# --------------------------------------------------------------------------
VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
VGG_TAPS = (3, 8, 15, 22, 29)      # post-ReLU module indices (loss.py:71)


def vgg_conv_table() -> List[Tuple[int, int, int]]:
    """[(features index, cin, cout)] of the 13 convs of VGG16 features[0:30]."""
    out, idx, cin = [], 0, 3
    for v in VGG_CFG:
        if v == "M":
            idx += 1
        else:
            out.append((idx, cin, v))
            cin = v
            idx += 2
    return out


def normalize_(img: Tensor, lo: float = -1024.0, hi: float = 3072.0) -> Tensor:
    return (img - lo) / (hi - lo)                                 # loss.py:20-22


def vgg_taps(vgg: Dict[str, Tensor], x3: Tensor) -> List[Tensor]:
    """Vgg16FeatureExtractor.forward: 5 post-ReLU taps; the ImageNet mean/std
    normalisation is commented out in the reference (loss.py:81-87)."""
    feats, idx, h = [], 0, x3
    for v in VGG_CFG:
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            idx += 1
        else:
            w = vgg[f"{idx}.weight"] if idx == 0 else _r(vgg[f"{idx}.weight"])
            h = F.relu(F.conv2d(h, w, vgg[f"{idx}.bias"], 1, 1))
            idx += 2
            if idx - 1 in VGG_TAPS:
                feats.append(h)          # taps are compared in fp32
            h = _r(h)
    return feats


def d_loss(pos: Tensor, neg: Tensor) -> Tensor:
    """DLoss: hinge applied to the batch means (loss.py:141-142)."""
    return F.relu(1 - pos.mean()) + F.relu(1 + neg.mean())


def g_loss(neg: Tensor) -> Tensor:
    return 1.0 - neg.mean()                                       # loss.py:154-155


def compound_loss(vgg: Dict[str, Tensor], pred: Tensor, gt: Tensor, d_out: Tensor,
                  vgg_weight=0.3, ssim_weight=0.5, gen_weight=0.2):
    """CompoundLoss.forward (loss.py:177-197).  Returns (loss, vgg_part, dssim_part, gen_part)."""
    pn, gn = normalize_(pred), normalize_(gt)
    fp = vgg_taps(vgg, torch.cat([pn, pn, pn], 1))
    with torch.no_grad():
        fg = vgg_taps(vgg, torch.cat([gn, gn, gn], 1))
    vl = sum(F.mse_loss(a, b) for a, b in zip(fp, fg)) / len(fp)
    sl = dssim_loss(gt, pred)
    gl = g_loss(d_out)
    return vgg_weight * vl + ssim_weight * sl + gen_weight * gl, vl, sl, gl


# --------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults as used at train.py:104-105)
# --------------------------------------------------------------------------
def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                b1: float = 0.01, b2: float = 0.99, eps: float = 1e-8):
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


# --------------------------------------------------------------------------
# training step (src/train.py:153-177) and inference (src/test.py:125-144)
# --------------------------------------------------------------------------
_NON_TRAINABLE = ("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v")


def trainable(name: str) -> bool:
    return not name.endswith(_NON_TRAINABLE)


class TrainState:
    """G and D state dicts (reference key names) plus Adam moments."""

    def __init__(self, g_sd: Dict[str, Tensor], d_sd: Dict[str, Tensor], vgg: Dict[str, Tensor], lr: float = 1e-4):
        self.g = {k: v.clone() for k, v in g_sd.items()}
        self.d = {k: v.clone() for k, v in d_sd.items()}
        self.vgg = vgg
        self.lr = lr
        self.g_m = {k: torch.zeros_like(v) for k, v in self.g.items() if trainable(k)}
        self.g_v = {k: torch.zeros_like(v) for k, v in self.g.items() if trainable(k)}
        self.d_m = {k: torch.zeros_like(v) for k, v in self.d.items() if trainable(k)}
        self.d_v = {k: torch.zeros_like(v) for k, v in self.d.items() if trainable(k)}
        self.g_step = 0
        self.d_step = 0


def train_step(st: TrainState, x: Tensor, y: Tensor, want_grads: bool = False, fresh_d: bool = False,
               shards: int = 1) -> Dict[str, object]:
    """One iteration of train.py:153-177 under the one well-defined reading of it:

      pred = G(x)                                   (train.py:153)
      2 x { D(cat[y, pred]); DLoss; backward; D Adam step }   (156-165)
      gloss = CompoundLoss(pred, y, pred_neg of the LAST D forward)  (172)
      G gradient = d gloss / d G-params, taken through D as it was at that
      last forward (weights and sigma before the final D Adam step);
      G Adam step.                                  (176-177)

    The verbatim script raises on torch >= 1.5 at gloss.backward (in-place D
    update between forward and backward); the forward values below are the
    script's.  Work whose result the script discards (D-loss gradient into G,
    VGG weight gradients) is not computed.

    fresh_d: the alternative the script's author left commented out at
    train.py:171 (`D_gen = Dnet(pred)`): pred_neg for the generator loss comes
    from a new training-mode forward of the discriminator on pred after BOTH
    of its updates (so the spectral norms run one more power iteration), and
    the G gradient goes through that discriminator.

    shards > 1: the data-parallel form of the same step, W equal shards of the
    batch on W replicas.  The generator's train-mode BatchNorm sees one shard
    at a time (per-replica statistics: what the nn.DataParallel commented out
    at train.py:95-96 would do); nothing else couples the samples of a batch
    (the discriminator has no BatchNorm, attention is per image), and every
    loss is a mean over the global batch - DLoss / GLoss of the GLOBAL means
    (loss.py:141-142,154-155) - so the gradients below equal the average over
    replicas of the per-replica gradients.  Running statistics: shard 0's.
    """
    gp = {k: (v.clone().requires_grad_(True) if trainable(k) else v.clone()) for k, v in st.g.items()}
    PG = Params(gp)
    if shards > 1:
        assert x.size(0) % shards == 0
        n = x.size(0) // shards
        preds = []
        for s in range(shards):      # shared parameter leaves, per-replica buffers (replica 0's are kept)
            Ps = PG if s == 0 else Params({k: (v if trainable(k) else v.clone()) for k, v in st.g.items()} | {
                k: gp[k] for k in gp if trainable(k)})
            preds.append(rigan_forward(Ps, x[s * n:(s + 1) * n], training=True))
        pred = torch.cat(preds, 0)
    else:
        pred = rigan_forward(PG, x, training=True)
    out: Dict[str, object] = {}
    dlosses = []
    g_grads = None

    def g_update_terms(neg):
        gl, vl, sl, genl = compound_loss(st.vgg, pred, y, neg)
        gnames = [k for k in gp if trainable(k)]
        if want_grads:      # d gloss / d pred and its three parts (diagnostics for the tests)
            for nm, term in (("dpred_vgg", 0.3 * vl), ("dpred_dssim", 0.5 * sl), ("dpred_gen", 0.2 * genl)):
                out[nm] = torch.autograd.grad(term, pred, retain_graph=True)[0].detach()
            out["dpred"] = out["dpred_vgg"] + out["dpred_dssim"] + out["dpred_gen"]
        gg = dict(zip(gnames, torch.autograd.grad(gl, [gp[k] for k in gnames])))
        out.update(gloss=gl.item(), vgg=vl.item(), dssim=sl.item(), gen=genl.item())
        return gg

    for it in range(2):
        dp = {k: (v.clone().requires_grad_(True) if trainable(k) else v.clone()) for k, v in st.d.items()}
        PD = Params(dp)
        last = it == 1 and not fresh_d
        d_in = torch.cat([y, pred if last else pred.detach()], 0)
        d_out = discriminator_forward(PD, d_in, training=True)
        pos, neg = torch.chunk(d_out, 2, 0)
        dl = d_loss(pos, neg)
        names = [k for k in dp if trainable(k)]
        grads = torch.autograd.grad(dl, [dp[k] for k in names], retain_graph=last, allow_unused=True)
        if last:
            g_grads = g_update_terms(neg)
        st.d_step += 1
        with torch.no_grad():
            for k, g in zip(names, grads):
                g = torch.zeros_like(dp[k]) if g is None else g
                st.d[k], st.d_m[k], st.d_v[k] = adam_update(st.d[k], g, st.d_m[k], st.d_v[k], st.d_step, 4 * st.lr)
            for k in dp:
                if not trainable(k):
                    st.d[k] = dp[k].detach()
        dlosses.append(dl.item())
        if want_grads:
            out[f"d_grads_{it}"] = {k: (g.detach() if g is not None else None) for k, g in zip(names, grads)}
    if fresh_d:
        dp = {k: v.clone() for k, v in st.d.items()}
        PD = Params(dp)
        neg = discriminator_forward(PD, pred, training=True)
        g_grads = g_update_terms(neg)
        for k in dp:
            if not trainable(k):
                st.d[k] = dp[k].detach()
    st.g_step += 1
    with torch.no_grad():
        for k, g in g_grads.items():
            st.g[k], st.g_m[k], st.g_v[k] = adam_update(st.g[k], g, st.g_m[k], st.g_v[k], st.g_step, st.lr)
        for k in gp:
            if not trainable(k):
                st.g[k] = gp[k].detach()
    out.update(dloss1=dlosses[0], dloss2=dlosses[1], pred=pred.detach())
    if want_grads:
        out["g_grads"] = {k: g.detach() for k, g in g_grads.items()}
    return out


def infer_slice(g_sd: Dict[str, Tensor], x: Tensor, y: Tensor, lo=-160.0, hi=240.0):
    """test.py:125-144 for one batch: G forward (train-mode BN, as the script
    never calls eval()), clamp x, y, pred to [lo, hi], per-slice compute_measure
    with data_range = hi - lo.  Returns (pred, list of (orig 3-tuple, pred 3-tuple))."""
    P = Params({k: v.clone() for k, v in g_sd.items()})
    with torch.no_grad():
        pred = rigan_forward(P, x, training=True)
    res = []
    for i in range(x.size(0)):
        xi, yi, pi = trunc(x[i, 0], lo, hi), trunc(y[i, 0], lo, hi), trunc(pred[i, 0], lo, hi)
        res.append(compute_measure(xi, yi, pi, hi - lo))
    return pred, res, P.sd


# ---------------------------------------------------------------------------
# training patches (src/loader.py)
# ---------------------------------------------------------------------------
def get_patch(full_input_img, full_target_img, patch_n: int, patch_size: int):
    """src/loader.py:60-73 on numpy arrays: patch_n crops at positions drawn from numpy's global generator,
    top then left (loader.py:67-68), the same position in both images.  Pinned by tests/golden/loader.npz
    (made by calling the reference's get_patch / get_loader on seeded synthetic slices)."""
    import numpy as np
    assert full_input_img.shape == full_target_img.shape
    h, w = full_input_img.shape
    xs, ys = [], []
    for _ in range(patch_n):
        top = np.random.randint(0, h - patch_size)
        left = np.random.randint(0, w - patch_size)
        xs.append(full_input_img[top:top + patch_size, left:left + patch_size])
        ys.append(full_target_img[top:top + patch_size, left:left + patch_size])
    return np.array(xs), np.array(ys)


def loader_batches(inputs, targets, patch_n: int, patch_size: int, batch_size: int):
    """What iterating get_loader(...) yields (src/loader.py:41-57,76-82: in order, no shuffle, default collate):
    (x, y) arrays of shape (batch, patch_n, P, P), the last batch possibly smaller."""
    import numpy as np
    for first in range(0, len(inputs), batch_size):
        xs, ys = [], []
        for i in range(first, min(first + batch_size, len(inputs))):
            a, b = get_patch(inputs[i], targets[i], patch_n, patch_size)
            xs.append(a); ys.append(b)
        yield np.stack(xs), np.stack(ys)


def device_patch_offsets(seed: int, first_draw: int, n: int, h: int, w: int, patch_size: int):
    """The crop positions the device generator of ried_patch_crop uses when no offsets are supplied: splitmix64 of
    (seed, draw), top from the high and left from the low 32 bits, each scaled to [0, h-P) / [0, w-P) by a
    multiply-shift.  (The product's own definition, restated with Python integers; no reference counterpart.)"""
    import numpy as np
    M = (1 << 64) - 1
    out = np.empty((n, 2), dtype=np.int32)
    for j in range(n):
        z = (seed + (first_draw + j + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        out[j, 0] = ((z >> 32) * (h - patch_size)) >> 32
        out[j, 1] = ((z & 0xffffffff) * (w - patch_size)) >> 32
    return out


# ---------------------------------------------------------------------------
# MPL (src/loss.py:89-110)
# ---------------------------------------------------------------------------
def mpl_loss(vgg: Dict[str, Tensor], target: Tensor, prediction: Tensor) -> Tensor:
    """src/loss.py:96-110: per tap (modules 3, 8, 15, 22, 29 of VGG16 features on cat([img]*3) of the RAW images,
    loss.py:36 applies no normalisation) the channel L2 norm of the feature difference / (H*W), mean over batch and
    positions, summed over taps.  Parity unpinned: src/loss.py is not importable here (torchvision is absent), so
    this follows the text of the reference."""
    ft = vgg_taps(vgg, torch.cat([target, target, target], 1))
    fp = vgg_taps(vgg, torch.cat([prediction, prediction, prediction], 1))
    total = torch.zeros((), dtype=torch.float32)
    for a, b in zip(ft, fp):
        _, _, h, w = a.shape
        total = total + (torch.linalg.norm(a - b, dim=1, ord=2) / float(h * w)).mean()
    return total


# --------------------------------------------------------------------------
# legacy networks (src/networks.py:460-556); pinned by tests/golden/legacy_*.npz
# --------------------------------------------------------------------------
def _legacy_act(y: Tensor, act: str) -> Tensor:
    if act == "relu":
        return F.relu(y)
    if act == "leaky":
        return F.leaky_relu(y, 0.2)
    if act == "sigmoid":
        return torch.sigmoid(y)
    return y


def legacy_conv_bn(P: Params, pre: str, x: Tensor, cin: int, cout: int, k: int, stride: int, training: bool,
                   act: str = "relu", deconv: bool = False, round_out: bool = True) -> Tensor:
    """Sequential(Conv2d / ConvTranspose2d with bias, BatchNorm2d[, activation]) as at src/networks.py:468-484,
    531-542: keys <pre>.0.* (conv) and <pre>.1.* (BN).
    With the bf16 emulation on and in train mode the conv runs without its bias, as the HIP path does: the
    normalised output does not depend on it, only running_mean does, which gets momentum * bias afterwards."""
    wshape = (cin, cout, k, k) if deconv else (cout, cin, k, k)
    w = _r(P.get(pre + ".0.weight", wshape))
    b = P.get(pre + ".0.bias", (cout,))
    conv = F.conv_transpose2d if deconv else F.conv2d
    drop = EMULATE_BF16 and training
    y = conv(x, w, None if drop else b, stride, 1)
    y = _bn_relu(P, pre + ".1", y, cout, training, relu=False, round_out=False)
    if drop:
        P.set(pre + ".1.running_mean", P.sd[pre + ".1.running_mean"] + BN_MOMENTUM * b.detach())
    y = _legacy_act(y, act)
    return _r(y) if round_out else y


def gnet_forward(P: Params, x: Tensor, training: bool = True, acts: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """GNet.forward (src/networks.py:487-521).  The one `residual` block is applied five times, so its BatchNorm
    running statistics advance five times per forward and its weights collect five gradient terms."""
    t = training
    a = {} if acts is None else acts
    w1, b1 = P.get("conv1.0.weight", (64, 1, 3, 3)), P.get("conv1.0.bias", (64,))
    a["conv1"] = c1 = _r(F.relu(F.conv2d(x, w1, b1, 1, 1)))          # fp32 weights on the raw image
    a["layer1"] = l1 = legacy_conv_bn(P, "layer1", c1, 64, 128, 3, 2, t)
    a["layer2"] = s = legacy_conv_bn(P, "layer2", l1, 128, 256, 3, 2, t)
    for _ in range(5):
        w_a, b_a = _r(P.get("residual.0.weight", (256, 256, 3, 3))), P.get("residual.0.bias", (256,))
        w_b, b_b = _r(P.get("residual.3.weight", (256, 256, 3, 3))), P.get("residual.3.bias", (256,))
        drop = EMULATE_BF16 and t
        h = _bn_relu(P, "residual.1", F.conv2d(s, w_a, None if drop else b_a, 1, 1), 256, t)
        r = _bn_relu(P, "residual.4", F.conv2d(h, w_b, None if drop else b_b, 1, 1), 256, t, relu=False,
                     round_out=False)
        if drop:
            P.set("residual.1.running_mean", P.sd["residual.1.running_mean"] + BN_MOMENTUM * b_a.detach())
            P.set("residual.4.running_mean", P.sd["residual.4.running_mean"] + BN_MOMENTUM * b_b.detach())
        a["residual"] = r
        s = _r(r + s)
    a["deconv1"] = d1 = legacy_conv_bn(P, "deconv1", s, 256, 256, 4, 2, t, deconv=True)
    a["deconv2"] = d2 = legacy_conv_bn(P, "deconv2", torch.cat([l1, d1], 1), 384, 128, 4, 2, t, deconv=True)
    w2, b2 = P.get("conv2.weight", (1, 192, 3, 3)), P.get("conv2.bias", (1,))
    a["out"] = out = F.conv2d(torch.cat([c1, d2], 1), w2, b2, 1, 1)   # fp32 weights, fp32 result
    return out


def dnet_forward(P: Params, x: Tensor, training: bool = True, acts: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """DNet.forward (src/networks.py:544-556): (B,1,H,W) -> (B,1024,H/32,W/32) in (0, 1)."""
    a = {} if acts is None else acts
    w1, b1 = P.get("conv1.0.weight", (64, 1, 3, 3)), P.get("conv1.0.bias", (64,))
    a["conv1"] = h = _r(F.leaky_relu(F.conv2d(x, w1, b1, 2, 1), 0.2))
    c = 64
    for i in range(2, 6):
        a[f"conv{i}"] = h = legacy_conv_bn(P, f"conv{i}", h, c, 2 * c, 3, 2, training,
                                           act="sigmoid" if i == 5 else "leaky")
        c *= 2
    return h


def gnet_spec() -> List[Tuple[str, Tuple[int, ...], str]]:
    P = Params(spec_mode=True)
    with torch.no_grad():
        gnet_forward(P, torch.zeros(2, 1, 8, 8), training=True)
    return P.spec


def dnet_spec() -> List[Tuple[str, Tuple[int, ...], str]]:
    P = Params(spec_mode=True)
    with torch.no_grad():
        dnet_forward(P, torch.zeros(2, 1, 32, 32), training=True)
    return P.spec


# --------------------------------------------------------------------------
# DICOM -> HU (src/prep.py:46-59).  The decode itself belongs to pydicom (requirements.txt pins pydicom 2.2.2),
# which is absent here: parity of the file decode is unpinned; the arithmetic below restates prep.py line by line.
# --------------------------------------------------------------------------
def get_pixels_hu(pixel_arrays, slopes, intercepts):
    """pixel_arrays: list of (H, W) stored-value arrays (uint16 or int16), slopes / intercepts: RescaleSlope /
    RescaleIntercept per slice -> int16 (n, H, W)."""
    import numpy as np
    image = np.stack(pixel_arrays)
    image = image.astype(np.int16)                               # prep.py:49
    image[image == -2000] = 0                                    # prep.py:50
    for i in range(len(pixel_arrays)):
        if slopes[i] != 1:                                       # prep.py:54-56
            image[i] = slopes[i] * image[i].astype(np.float64)
            image[i] = image[i].astype(np.int16)
        image[i] += np.int16(intercepts[i])                      # prep.py:57
    return np.array(image, dtype=np.int16)
