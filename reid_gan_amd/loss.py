"""Drop-in loss API of the reference's src/loss.py on the MI355X kernels.

Same names and call signatures: normalize_, gaussian, create_window, compute_SSIM, Vgg16FeatureExtractor, SSIM,
DLoss, GLoss, CompoundLoss, to_cuda.  Every forward is a HIP kernel sequence wrapped in a
torch.autograd.Function whose backward is the matching kernel sequence; there is no CPU path.

VGG16 weights: the reference downloads torchvision's ImageNet weights (src/loss.py:95,167).  That is not possible
offline and torchvision is not a dependency here, so the weights come from, in order: the `vgg_state` argument,
the file named by $RIED_VGG16_STATE (a state dict with torchvision's keys `features.<i>.weight|bias`, loaded
with weights_only=True), or seeded synthetic weights (with a warning).
"""
from __future__ import annotations

import os
import warnings
from math import exp
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from .engine import Vgg
from .synth import synth_vgg

__all__ = ["normalize_", "gaussian", "create_window", "get_feature_layer_vgg16", "compute_SSIM",
           "Vgg16FeatureExtractor", "MPL", "SSIM", "DLoss", "GLoss", "CompoundLoss", "to_cuda", "load_vgg16_state"]

F32 = torch.float32
cuda_is_present = torch.cuda.is_available()


def to_cuda(data):
    return data.cuda() if torch.cuda.is_available() else data


def normalize_(image, MIN_B=-1024.0, MAX_B=3072.0):
    """(image - MIN_B) / (MAX_B - MIN_B) (reference: src/loss.py:20-22).  Inside CompoundLoss this affine is folded
    into the first VGG layer's load; as a free function it is plain tensor arithmetic on the caller's device."""
    return (image - MIN_B) / (MAX_B - MIN_B)


def gaussian(window_size, sigma):
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return to_cuda(w2.expand(channel, 1, window_size, window_size).contiguous())


def _as_images(t: torch.Tensor) -> torch.Tensor:
    """fp32 contiguous (N, H, W) device view of a 2-d / 4-d one-channel tensor."""
    if t.dim() == 2:
        t = t.reshape(1, *t.shape)
    elif t.dim() == 4:
        if t.shape[1] != 1:
            raise NotImplementedError("SSIM kernels take one-channel images")
        t = t.reshape(t.shape[0], t.shape[2], t.shape[3])
    elif t.dim() != 3:
        raise ValueError("expected a 2-d, 3-d or 4-d tensor")
    return t.to(device="cuda", dtype=F32).contiguous()


class _SsimFn(torch.autograd.Function):
    """mean (or per-image mean) of the SSIM map of (img1, img2); differentiable in both images."""

    @staticmethod
    def forward(ctx, img1, img2, c1, c2, size_average):
        a, b = _as_images(img2), _as_images(img1)        # the kernel differentiates its first image
        n, h, w = a.shape
        sums = torch.zeros(n, 4, dtype=F32, device="cuda")
        need = img1.requires_grad or img2.requires_grad
        gmap = torch.empty(3, n, h, w, dtype=F32, device="cuda") if need else None
        ops.ssim_fwd(a, None, b, c1, c2, sums, gmap=gmap)
        ctx.save_for_backward(a, b, gmap if need else torch.empty(0))
        ctx.meta = (c1, c2, size_average, img1.shape, img2.shape, img1.requires_grad, img2.requires_grad)
        per = sums[:, 0] / (h * w)
        return per.mean() if size_average else per

    @staticmethod
    def backward(ctx, go):
        a, b, gmap = ctx.saved_tensors
        c1, c2, size_average, s1, s2, need1, need2 = ctx.meta
        n, h, w = a.shape
        if not size_average:
            raise NotImplementedError("gradient of the per-image SSIM is not provided")
        scale = go.reshape(1).to(F32) / (n * h * w)
        g1 = g2 = None
        if need2:
            g2 = torch.empty_like(a)
            ops.ssim_bwd(a, b, gmap, 1.0, g2, scale_dev=scale)
            g2 = g2.reshape(s2)
        if need1:            # SSIM is symmetric: swap the roles
            sums = torch.zeros(n, 4, dtype=F32, device="cuda")
            gm = torch.empty_like(gmap)
            ops.ssim_fwd(b, None, a, c1, c2, sums, gmap=gm)
            g1 = torch.empty_like(b)
            ops.ssim_bwd(b, a, gm, 1.0, g1, scale_dev=scale)
            g1 = g1.reshape(s1)
        return g1, g2, None, None, None


def compute_SSIM(img1, img2, window_size, channel, size_average=True):
    """Gaussian-window SSIM with C1 = 0.01**2, C2 = 0.03**2 (reference: src/loss.py:42-67)."""
    if window_size != 11 or channel != 1:
        raise NotImplementedError("the fused SSIM kernel implements window_size=11, channel=1")
    ops.require_gpu()
    return _SsimFn.apply(img1, img2, 0.01 ** 2, 0.03 ** 2, size_average)


class SSIM(nn.Module):
    """Dissimilarity loss (1 - SSIM)/2, called as forward(target, pred) (reference: src/loss.py:112-130)."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average, self.channel = window_size, size_average, 1
        self.window = create_window(window_size, self.channel)

    def forward(self, target, pred):
        return (1.0 - compute_SSIM(target, pred, self.window_size, self.channel, self.size_average)) / 2.0


class _MeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        x = t.to(device="cuda", dtype=F32).contiguous()
        out = torch.zeros(1, dtype=F32, device="cuda")
        ops.sum_f32(x, x.numel(), 1, 1.0 / x.numel(), out)
        ctx.meta = (t.shape, x.numel())
        return out.reshape(())

    @staticmethod
    def backward(ctx, go):
        shape, n = ctx.meta
        return (go.to(F32) / n).expand(shape)


class DLoss(nn.Module):
    """relu(1 - mean(pos)) + relu(1 + mean(neg)): hinge on the batch means (reference: src/loss.py:132-144)."""

    def __init__(self, weight=1):
        super().__init__()
        self.activation = nn.ReLU()
        self.weight = weight

    def forward(self, pos, neg):
        ops.require_gpu()
        return torch.clamp(1 - _MeanFn.apply(pos), min=0) + torch.clamp(1 + _MeanFn.apply(neg), min=0)


class GLoss(nn.Module):
    """1 - mean(neg) (reference: src/loss.py:146-155)."""

    def __init__(self, weight=1):
        super().__init__()
        self.weight = weight

    def forward(self, neg):
        ops.require_gpu()
        return 1.0 - _MeanFn.apply(neg)


def load_vgg16_state(vgg_state: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """{'<features idx>.weight|bias': tensor} for the 13 convs of VGG16 features[0:30]."""
    if vgg_state is None and os.environ.get("RIED_VGG16_STATE"):
        vgg_state = torch.load(os.environ["RIED_VGG16_STATE"], map_location="cpu", weights_only=True)
    if vgg_state is None:
        warnings.warn("no VGG16 weights given (vgg_state / $RIED_VGG16_STATE): using seeded synthetic weights; "
                      "perceptual-loss values will not match an ImageNet-pretrained VGG16")
        return synth_vgg()
    out = {}
    for k, v in vgg_state.items():
        k2 = k
        for pre in ("model.model.", "model.features.", "features."):
            if k2.startswith(pre):
                k2 = k2[len(pre):]
        parts = k2.split(".")
        if len(parts) == 2 and parts[0].isdigit() and int(parts[0]) < 30:
            out[k2] = v.detach().to(F32)
    if len(out) != 26:
        raise ValueError(f"expected 13 conv weights and biases of VGG16 features, found {len(out)} tensors")
    return out


class Vgg16FeatureExtractor(nn.Module):
    """The five post-ReLU taps (modules 3, 8, 15, 22, 29) of VGG16 features (reference: src/loss.py:69-87).
    forward(x) takes the reference's (B, 3, H, W) input whose three channels are identical copies (the only way
    the reference calls it, src/loss.py:181-182), already normalised; it returns the taps as NCHW fp32 tensors.
    No gradient: inside CompoundLoss the fused path below is used instead."""

    def __init__(self, layers=[3, 8, 15, 22, 29], pretrained=False, progress=True, vgg_state=None, **kwargs):
        super().__init__()
        if list(layers) != [3, 8, 15, 22, 29]:
            raise NotImplementedError("the HIP plan taps modules 3, 8, 15, 22, 29")
        self.layers = layers
        self.weights = load_vgg16_state(vgg_state)
        self._plans = {}

    def forward(self, x):
        ops.require_gpu()
        if x.shape[1] == 3:
            if not (torch.equal(x[:, 0], x[:, 1]) and torch.equal(x[:, 0], x[:, 2])):
                raise NotImplementedError("Vgg16FeatureExtractor: the three input channels must be identical")
            x = x[:, :1]
        B, _, H, W = x.shape
        if B % 2:
            raise NotImplementedError("Vgg16FeatureExtractor: batch must be even")
        key = (B, H, W)
        V = self._plans.get(key)
        if V is None:
            V = self._plans[key] = Vgg(self.weights, B // 2, H, W, grads=False, device=x.device)
        # the plan applies normalize_ on load; undo it for already-normalised input
        V.forward((x.to(F32) * 4096.0 - 1024.0).contiguous())
        return [t32.permute(0, 3, 1, 2).contiguous() for _, t32 in V.taps]


_TAP_OF_LAYER = {3: 0, 8: 1, 15: 2, 22: 3, 29: 4}


def get_feature_layer_vgg16(image, layer, model):
    """Output of module `layer` of VGG16 features on cat([image]*3, 1) (reference: src/loss.py:35-40).  image:
    (B, 1, H, W), fed as it is (the reference applies no normalisation here); layer: one of the post-ReLU taps 3,
    8, 15, 22, 29; model: an MPL or Vgg16FeatureExtractor of this package (the owner of the VGG16 weights).
    Returns NCHW fp32, no gradient."""
    if layer not in _TAP_OF_LAYER:
        raise NotImplementedError("the HIP plan taps modules 3, 8, 15, 22, 29")
    ops.require_gpu()
    B, _, H, W = image.shape
    key = ("raw", B, H, W)
    V = model._plans.get(key)
    if V is None:
        if B % 2:
            raise NotImplementedError("get_feature_layer_vgg16: batch must be even")
        V = model._plans[key] = Vgg(model.weights, B // 2, H, W, grads=False, device=image.device, normalize=False)
    V.forward(image.detach().to(F32).contiguous())
    return V.taps[_TAP_OF_LAYER[layer]][1].permute(0, 3, 1, 2).contiguous()


class _MplFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, target, prediction):
        B, _, H, W = prediction.shape
        V = mod._plan(B, H, W, prediction.device)
        p = prediction.detach().to(F32).contiguous()
        t = target.detach().to(F32).contiguous()
        loss = torch.zeros(1, dtype=F32, device=p.device)
        dpred = torch.zeros_like(p)
        V.forward(torch.cat([p, t], 0))
        V.ctx.zero_backward()
        V.mpl_loss_and_seed(loss)
        if prediction.requires_grad:
            V.backward(dpred)
        ctx.save_for_backward(dpred)
        ctx.pshape = prediction.shape
        return loss[0].clone()

    @staticmethod
    def backward(ctx, go):
        (dpred,) = ctx.saved_tensors
        return None, None, (dpred * go.to(F32)).reshape(ctx.pshape)


class MPL(nn.Module):
    """The Multi Perceptual Loss (reference: src/loss.py:89-110): for each of the taps 3, 8, 15, 22, 29 of VGG16
    on the raw (not normalised) images, the channel-wise L2 norm of the feature difference divided by the tap's
    H*W, averaged over batch and positions; summed over the taps.  forward(target, prediction); the gradient
    flows to `prediction` (the reference's training script never asks for the target's)."""

    def __init__(self, vgg_state=None):
        super().__init__()
        self.weights = load_vgg16_state(vgg_state)
        self._plans = {}

    def _plan(self, B, H, W, device):
        key = (B, H, W)
        if key not in self._plans:
            self._plans[key] = Vgg(self.weights, B, H, W, grads=True, device=device, normalize=False)
        return self._plans[key]

    def forward(self, target, prediction):
        ops.require_gpu()
        return _MplFn.apply(self, target, prediction)


class _CompoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, pred, gt, d_out):
        B, _, H, W = pred.shape
        V = mod._plan(B, H, W, pred.device)
        sc = torch.zeros(16, dtype=F32, device=pred.device)
        p = pred.detach().to(F32).contiguous()
        g = gt.detach().to(F32).contiguous()
        dpred = torch.zeros_like(p)
        V.forward(torch.cat([p, g], 0))
        V.ctx.zero_backward()
        V.loss_and_seed(sc[5:], mod.vgg_weight)
        V.backward(dpred)
        sums = torch.zeros(B, 4, dtype=F32, device=pred.device)
        gmap = torch.empty(3, B, H, W, dtype=F32, device=pred.device)
        ops.ssim_fwd(p, None, g, 0.01 ** 2, 0.03 ** 2, sums, gmap=gmap)
        ops.sum_f32(sums, B, 4, 1.0 / (B * H * W), sc[6:])
        ops.ssim_bwd(p, g, gmap, -mod.ssim_weight * 0.5 / (B * H * W), dpred, accumulate=True)
        d = d_out.detach().to(F32).contiguous()
        ops.sum_f32(d, d.numel(), 1, 1.0 / d.numel(), sc[1:])
        ops.gloss_scalars(sc, mod.vgg_weight, mod.ssim_weight, mod.gen_weight)
        ctx.save_for_backward(dpred)
        ctx.meta = (d_out.shape, d.numel(), mod.gen_weight, pred.shape)
        mod.last_terms = sc          # [5] perceptual, [7] dssim, [8] generator term, [9] total (device tensor)
        return sc[9].clone()

    @staticmethod
    def backward(ctx, go):
        (dpred,) = ctx.saved_tensors
        dshape, n, wg, pshape = ctx.meta
        go = go.to(F32)
        return None, (dpred * go).reshape(pshape), None, (-(wg / n) * go).expand(dshape)


class CompoundLoss(nn.Module):
    """vgg_weight * mean_5 MSE(VGG taps) + ssim_weight * DSSIM + gen_weight * (1 - mean(D(pred)))
    (reference: src/loss.py:157-197), as one fused kernel sequence with its backward."""

    def __init__(self, blocks=[1, 2, 3, 4, 5], vgg_weight=0.3, ssim_weight=0.5, gen_weight=0.2, vgg_state=None):
        super().__init__()
        if list(blocks) != [1, 2, 3, 4, 5]:
            raise NotImplementedError("CompoundLoss: all five taps are used")
        self.vgg_weight, self.ssim_weight, self.gen_weight = vgg_weight, ssim_weight, gen_weight
        self.blocks = blocks
        self.vgg_weights = load_vgg16_state(vgg_state)
        self._plans = {}
        self.last_terms = None

    def _plan(self, B, H, W, device):
        key = (B, H, W)
        if key not in self._plans:
            self._plans[key] = Vgg(self.vgg_weights, B, H, W, grads=True, device=device)
        return self._plans[key]

    def forward(self, pred, ground_truth, discriminator_out):
        ops.require_gpu()
        return _CompoundFn.apply(self, pred, ground_truth, discriminator_out)
