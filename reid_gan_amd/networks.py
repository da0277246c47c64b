"""Drop-in GNet / DNet of the reference's src/networks.py:460-556 on the MI355X kernels.

These are the reference's older generator and discriminator: importable through `from networks import *`
(src/train.py:20) but never built by its train.py / test.py.  Same class names, parameter names (state_dict keys)
and default initialisation; forward and backward run static HIP plans assembled from the kernels the RIGAN path
uses, plus the three of csrc/legacy.hip.  The remaining classes of that file (`block`, `ResNet`, `Inception_3`, ...)
are unfinished scaffolds there (Inception_3.forward ends in quit(), src/networks.py:456) and have no counterpart.
There is no eager / CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import engine, ops
from .engine import ConvBN, Ctx, Identity, Seq, Slot, State
from .rinet import F32, _NetFn, _PlanModule

__all__ = ["GNet", "DNet"]


class _ConvC1Act:
    """Conv2d(1, 64, 3, stride, padding=1) with bias on the fp32 image, then ReLU or LeakyReLU(0.2)
    (src/networks.py:466-467, 529-530)."""

    def __init__(self, ctx: Ctx, key: str, stride: int, act: int, N, H, W, out: Slot):
        self.ctx, self.key, self.stride, self.act, self.out = ctx, key, stride, act, out
        self.slope = 0.0 if act == ops.ACT_RELU else 0.2
        _, Ho, Wo, C = out.shape
        assert C == 64 and (Ho, Wo) == ((H - 1) // stride + 1, (W - 1) // stride + 1)
        if ctx.grads:
            ctx.need_tmp(N * Ho * Wo * C)
            ctx.need_f32(ops.colsum_scratch_floats(C))
            ctx.need_f32(ops.conv_c1_wgrad_scratch_floats(C))
        self.x: Optional[torch.Tensor] = None

    def forward(self, x: torch.Tensor):
        st = self.ctx.st
        self.x = x
        ops.conv_c1_fwd(x, st.p[self.key + ".weight"], self.out.v, self.stride, bias=st.p[self.key + ".bias"],
                        act=self.act)

    def backward(self, dimg: Optional[torch.Tensor] = None):
        """Reads out.g; parameter gradients add onto the state's; dimg: fp32 image gradient, overwritten."""
        ctx, st = self.ctx, self.ctx.st
        N, Ho, Wo, C = self.out.shape
        dpre = ctx.tmp_view(N, Ho, Wo, C)
        ops.ew(ops.EW_ACTBWD, self.out.g, dpre, y=self.out.v, slope=self.slope)
        ops.colsum(dpre, st.g[self.key + ".bias"], ctx.f32)
        ops.conv_c1_wgrad(self.x, dpre, self.stride, st.g[self.key + ".weight"], ctx.f32)
        if dimg is not None:
            if self.stride == 1:
                ops.conv_c1_dgrad(dpre, st.p[self.key + ".weight"], dimg)
            else:
                ops.conv_c1_dgrad_s2(dpre, st.p[self.key + ".weight"], dimg)


class GNetPlan:
    """GNet (src/networks.py:460-521) for a fixed (N, H, W); x and the output are fp32 (N,1,H,W)."""
    USES = 5          # the one residual block object is applied five times (src/networks.py:495-513)

    def __init__(self, state: State, N: int, H: int, W: int, grads=True, training=True, device="cuda"):
        assert H % 4 == 0 and W % 4 == 0, "GNet needs H, W divisible by 4 (skip connections must line up)"
        self.st = state
        self.ctx = ctx = Ctx(state, device, grads, training)
        self.N, self.H, self.W = N, H, W
        h2, w2, h4, w4 = H // 2, W // 2, H // 4, W // 4
        self.cat2 = Slot(ctx.act(N, H, W, 192))          # cat(conv1 (64), deconv2 (128))    src/networks.py:519
        self.cat1 = Slot(ctx.act(N, h2, w2, 384))        # cat(layer1 (128), deconv1 (256))  src/networks.py:517
        c1, l1 = self.cat2.sub(0, 64), self.cat1.sub(0, 128)
        self.conv1 = _ConvC1Act(ctx, "conv1.0", 1, ops.ACT_RELU, N, H, W, c1)
        body = Seq()

        def cbn(name, cin, cout, k, s, xin, out, **kw):
            return body.add(ConvBN(ctx, name, cin, cout, k, s, 1, xin, out, conv_key=name + ".0", bn_key=name + ".1",
                                   bias=True, **kw))
        cbn("layer1", 64, 128, 3, 2, c1, l1)
        s = Slot(ctx.act(N, h4, w4, 256))
        cbn("layer2", 128, 256, 3, 2, l1, s)
        # Each use of the shared block computes its own weight gradient; uses 2..5 write theirs to private
        # buffers that backward() adds onto the first one's (the slab reductions of one plan run concurrently).
        self.extra_wgrads = []
        for i in range(self.USES):
            mid, nxt = Slot(ctx.act(N, h4, w4, 256)), Slot(ctx.act(N, h4, w4, 256))
            wa = wb = None
            if grads and i > 0:
                wa, wb = (torch.zeros(256 * 256 * 9, dtype=F32, device=device) for _ in range(2))
                self.extra_wgrads += [("residual.0.weight", wa), ("residual.3.weight", wb)]
            body.add(Identity(s, nxt))                                                   # res.add(x)
            body.add(ConvBN(ctx, "residual", 256, 256, 3, 1, 1, s, mid, conv_key="residual.0", bn_key="residual.1",
                            bias=True, shared=True, wgrad_out=wa))
            body.add(ConvBN(ctx, "residual", 256, 256, 3, 1, 1, mid, nxt, accumulate=True, conv_key="residual.3",
                            bn_key="residual.4", bias=True, act=ops.ACT_NONE, shared=True, wgrad_out=wb))
            s = nxt
        cbn("deconv1", 256, 256, 4, 2, s, self.cat1.sub(128, 384), deconv=True)
        cbn("deconv2", 384, 128, 4, 2, self.cat1, self.cat2.sub(64, 192), deconv=True)
        self.body = body
        self.out = torch.empty(N, 1, H, W, dtype=F32, device=device)
        ctx.finalize()
        self.x = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape == (self.N, 1, self.H, self.W) and x.dtype == F32 and x.is_contiguous()
        self.x = x
        self.ctx.begin_forward()
        self.conv1.forward(x)
        self.body.forward()
        ops.conv3_to1_fwd(self.cat2.v, self.st.p["conv2.weight"], self.st.p["conv2.bias"], self.out)
        return self.out

    def backward(self, dout: torch.Tensor):
        """dout: fp32 (N,1,H,W).  Fills state.g (which must be zero on entry)."""
        assert dout.shape == self.out.shape and dout.dtype == F32 and dout.is_contiguous()
        st = self.st
        self.ctx.zero_backward()
        ops.conv3_to1_dgrad(dout, st.p["conv2.weight"], self.cat2.g)
        ops.conv3_to1_wgrad(self.cat2.v, dout, st.g["conv2.weight"])
        ops.sum_f32(dout, dout.numel(), 1, 1.0, st.g["conv2.bias"])
        self.body.backward()
        self.conv1.backward()
        self.ctx.end_backward()
        for key, buf in self.extra_wgrads:
            ops.axpy_f32(st.g[key], buf, 1.0)


class DNetPlan:
    """DNet (src/networks.py:523-556) for a fixed (N, H, W): five stride-2 3x3 convs, BatchNorm from the second
    on, LeakyReLU(0.2) and a final Sigmoid.  Output: fp32 (N, 1024, H/32, W/32)."""

    def __init__(self, state: State, N: int, H: int, W: int, grads=True, training=True, device="cuda"):
        self.st = state
        self.ctx = ctx = Ctx(state, device, grads, training)
        self.N, self.H, self.W = N, H, W
        half = lambda v: (v - 1) // 2 + 1
        h, w = half(H), half(W)
        a = Slot(ctx.act(N, h, w, 64))
        self.conv1 = _ConvC1Act(ctx, "conv1.0", 2, ops.ACT_LEAKY, N, H, W, a)
        body = Seq()
        c = 64
        for i in range(2, 6):
            h, w = half(h), half(w)
            nxt = Slot(ctx.act(N, h, w, 2 * c))
            body.add(ConvBN(ctx, f"conv{i}", c, 2 * c, 3, 2, 1, a, nxt, conv_key=f"conv{i}.0", bn_key=f"conv{i}.1",
                            bias=True, act=ops.ACT_SIGMOID if i == 5 else ops.ACT_LEAKY))
            a, c = nxt, 2 * c
        self.body, self.a_out = body, a
        self.out = torch.empty(N, c, h, w, dtype=F32, device=device)
        self.dimg = torch.zeros(N, 1, H, W, dtype=F32, device=device) if grads else None
        ctx.finalize()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape == (self.N, 1, self.H, self.W) and x.dtype == F32 and x.is_contiguous()
        self.ctx.begin_forward()
        self.conv1.forward(x)
        self.body.forward()
        ops.nhwc_to_nchw(self.a_out.v, self.out)
        return self.out

    def backward(self, dout: torch.Tensor, need_dimg: bool = False) -> Optional[torch.Tensor]:
        assert dout.shape == self.out.shape and dout.dtype == F32 and dout.is_contiguous()
        self.ctx.zero_backward()
        ops.nchw_to_nhwc(dout, self.a_out.g)
        self.body.backward()
        self.conv1.backward(self.dimg if need_dimg else None)
        self.ctx.end_backward()
        return self.dimg if need_dimg else None


# --------------------------------------------------------------------------- the drop-in modules
class _LegacyNet(_PlanModule):
    _key_prefix = ""
    _plan_cls = None

    def _plans(self):
        if not hasattr(self, "_lplans"):
            object.__setattr__(self, "_lplans", {})
        return self._lplans

    def _net_forward(self, x):
        st = self._ensure_bound()
        key = (tuple(x.shape), self.training)
        plans = self._plans()
        if getattr(self, "_lplans_state", None) is not st:
            plans.clear()
            object.__setattr__(self, "_lplans_state", st)
        P = plans.get(key)
        if P is None:
            N, C, H, W = x.shape
            if C != 1:
                raise ValueError(f"{type(self).__name__} expects one-channel images")
            P = plans[key] = self._plan_cls(st, N, H, W, grads=True, training=self.training, device=x.device)
        P.x_keep = x.contiguous().to(F32)
        P.serial = getattr(P, "serial", 0) + 1        # (rinet._NetFn refuses the backward of a superseded forward)
        return P.forward(P.x_keep).clone()

    def forward(self, input):
        if input.dim() != 4:
            raise ValueError("expected a (N, 1, H, W) tensor")
        self._ensure_bound()
        return _NetFn.apply(self, input, *list(self.parameters()))


class GNet(_LegacyNet):
    """reference: src/networks.py:460-521."""
    _plan_cls = GNetPlan

    def __init__(self):
        super().__init__()
        self.input_channel, self.inter_channel = 1, 64
        c = self.inter_channel
        self.conv1 = nn.Sequential(nn.Conv2d(1, c, 3, 1, padding=1), nn.ReLU(inplace=True))
        self.layer1 = nn.Sequential(nn.Conv2d(c, c * 2, 3, 2, padding=1), nn.BatchNorm2d(c * 2), nn.ReLU(inplace=True))
        self.layer2 = nn.Sequential(nn.Conv2d(c * 2, c * 4, 3, 2, padding=1), nn.BatchNorm2d(c * 4),
                                    nn.ReLU(inplace=True))
        self.residual = nn.Sequential(nn.Conv2d(4 * c, 4 * c, 3, 1, padding=1), nn.BatchNorm2d(4 * c),
                                      nn.ReLU(inplace=True),
                                      nn.Conv2d(4 * c, 4 * c, 3, 1, padding=1), nn.BatchNorm2d(4 * c))
        self.deconv1 = nn.Sequential(nn.ConvTranspose2d(c * 4, c * 4, 4, 2, padding=1), nn.BatchNorm2d(c * 4),
                                     nn.ReLU(inplace=True))
        self.deconv2 = nn.Sequential(nn.ConvTranspose2d(c * 6, c * 2, 4, 2, padding=1), nn.BatchNorm2d(c * 2),
                                     nn.ReLU(inplace=True))
        self.conv2 = nn.Conv2d(c * 3, 1, 3, padding=1)

    def _net_backward(self, key, dy, need_dx):
        P = self._plans()[key]
        self._state.zero_grad()
        P.backward(dy)
        return None              # the network input is data


class DNet(_LegacyNet):
    """reference: src/networks.py:523-556."""
    _plan_cls = DNetPlan

    def __init__(self):
        super().__init__()
        self.input_channel, self.inter_channel = 1, 64
        c = self.inter_channel
        self.conv1 = nn.Sequential(nn.Conv2d(1, c, 3, 2, padding=1), nn.LeakyReLU(0.2, inplace=True))
        self.conv2 = nn.Sequential(nn.Conv2d(c, 2 * c, 3, 2, padding=1), nn.BatchNorm2d(2 * c),
                                   nn.LeakyReLU(0.2, inplace=True))
        self.conv3 = nn.Sequential(nn.Conv2d(2 * c, 4 * c, 3, 2, padding=1), nn.BatchNorm2d(4 * c),
                                   nn.LeakyReLU(0.2, inplace=True))
        self.conv4 = nn.Sequential(nn.Conv2d(4 * c, 8 * c, 3, 2, padding=1), nn.BatchNorm2d(8 * c),
                                   nn.LeakyReLU(0.2, inplace=True))
        self.conv5 = nn.Sequential(nn.Conv2d(8 * c, 16 * c, 3, 2, padding=1), nn.BatchNorm2d(16 * c), nn.Sigmoid())

    def _net_backward(self, key, dy, need_dx):
        P = self._plans()[key]
        self._state.zero_grad()
        dimg = P.backward(dy, need_dimg=need_dx)
        return dimg.clone() if need_dx else None
