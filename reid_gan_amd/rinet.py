"""Drop-in module API of the reference's src/rinet.py on the MI355X kernels.

Same class names, constructor arguments, parameter / buffer names (state_dict keys) and default initialisation
as the reference, so `from rinet import *` in its train.py / test.py can be pointed here and checkpoints load
either way.  The torch.nn layers created in the constructors are parameter containers only: forward and
backward run the static HIP plans of reid_gan_amd.engine through one torch.autograd.Function per module call.
There is no eager / CPU fallback: calling a module without a ROCm device raises.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import engine, ops
from .engine import Ctx, Seq, Slot, State

__all__ = ["ImageDiscriminator", "RIGAN", "SNConv2d", "Self_Attn", "BasicConv2d", "BasicDeconv2d", "InceptionA",
           "InceptionB", "InceptionC", "InceptionD", "InceptionE"]

F32 = torch.float32


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# --------------------------------------------------------------------------- binding modules to plans
def bind_state(module: nn.Module, key_prefix: str = "") -> State:
    """Flatten the module's parameters into one fp32 buffer (parameters become views of it, so in-place
    optimizer updates are seen by the kernels) and share its buffers."""
    sd = {}
    for k, p in module.named_parameters():
        sd[key_prefix + k] = p.detach()
    bufs = {key_prefix + k: b for k, b in module.named_buffers()}
    for k, b in bufs.items():
        sd[k] = b
    dev = next(module.parameters()).device
    st = State({k: v for k, v in sd.items()}, dev)
    for k, p in module.named_parameters():
        p.data = st.p[key_prefix + k]
    for k, b in bufs.items():          # buffers: use the module's own storage (running stats, u, v)
        st.b[k] = b
    st._bound_ptrs = [p.data_ptr() for p in module.parameters()]
    return st


def _check_serial(plan, serial: int, what: str):
    """A plan keeps the activations of its LAST forward only: a backward that belongs to an earlier forward of the
    same module and input shape would silently use the later call's activations, so it is refused."""
    if plan.serial != serial:
        raise RuntimeError(
            f"{what}: backward of forward call #{serial}, but the module has since run forward call #{plan.serial} "
            "with the same input shape; the plan holds one set of activations per shape. Run backward before the "
            "next forward of that shape, or batch the inputs into one call (as the reference's training loop does).")


class _PlanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        runner = module._runner(x)
        ctx.runner = runner
        ctx.need_dx = x.requires_grad
        y = runner.forward(x)
        ctx.serial = runner.serial
        return y

    @staticmethod
    def backward(ctx, dy):
        r = ctx.runner
        _check_serial(r, ctx.serial, type(r.module).__name__)
        dx = r.backward(dy.contiguous(), ctx.need_dx)
        grads = [r.state.g[r.key_prefix + k].clone() for k, _ in r.module.named_parameters()]
        return (None, dx, *grads)


class _Runner:
    """A plan for one (module, input shape, training flag)."""

    def __init__(self, module: "_PlanModule", shape, training: bool):
        self.module, self.key_prefix = module, module._key_prefix
        self.state = module._state
        N, C, H, W = shape
        self.ctx = ctx = Ctx(self.state, self.state.device, grads=True, training=training)
        self.a_in = Slot(ctx.act(N, H, W, C))
        self.plan, self.a_out = module._build(ctx, self.a_in)
        ctx.finalize()
        v = self.a_out.act.v
        self.y = torch.empty(N, self.a_out.C, v.H, v.W, dtype=F32, device=self.state.device)
        self.dx = torch.empty(N, C, H, W, dtype=F32, device=self.state.device)
        self.serial = 0           # forward calls so far

    def forward(self, x):
        self.serial += 1
        ops.nchw_to_nhwc(x.contiguous().to(F32), self.a_in.v)
        self.ctx.begin_forward()
        self.plan.forward()
        ops.nhwc_to_nchw(self.a_out.v, self.y)
        return self.y.clone()

    def backward(self, dy, need_dx):
        self.ctx.zero_backward()
        self.state.zero_grad()
        ops.nchw_to_nhwc(dy.to(F32), self.a_out.g)
        self.plan.backward()
        self.ctx.end_backward()
        if not need_dx:
            return None
        ops.nhwc_to_nchw(self.a_in.g, self.dx)
        return self.dx.clone()


class _PlanModule(nn.Module):
    """Base of every drop-in module: caches a State and one runner per input shape."""
    _key_prefix = "."          # block plans are built with prefix "", whose keys start with "."

    def _ensure_bound(self):
        st = getattr(self, "_state", None)
        ptrs = [p.data_ptr() for p in self.parameters()]
        if st is None or st._bound_ptrs != ptrs:
            ops.require_gpu()
            if not all(p.is_cuda for p in self.parameters()):
                raise RuntimeError(f"{type(self).__name__}: parameters must be on the ROCm device (call .cuda()); "
                                   "there is no CPU path")
            object.__setattr__(self, "_state", bind_state(self, self._key_prefix))
            object.__setattr__(self, "_runners", {})
        return self._state

    def _runner(self, x):
        self._ensure_bound()
        key = (tuple(x.shape), self.training)
        r = self._runners.get(key)
        if r is None:
            r = self._runners[key] = self._make_runner(tuple(x.shape), self.training)
        return r

    def _make_runner(self, shape, training):
        return _Runner(self, shape, training)

    def _build(self, ctx, xin: Slot):
        raise NotImplementedError

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError("expected a (N, C, H, W) tensor")
        self._ensure_bound()
        return _PlanFn.apply(self, x, *list(self.parameters()))


# --------------------------------------------------------------------------- basic layers
class BasicConv2d(_PlanModule):
    """Conv2d(no bias) -> BatchNorm2d -> ReLU (reference: src/rinet.py:183-193)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.activation = nn.ReLU(inplace=True)
        self._geom = (in_channels, out_channels, kernel_size, stride, padding)

    _deconv = False

    def _build(self, ctx, xin):
        cin, cout, k, s, p = self._geom
        N, H, W, _ = xin.shape
        kh, kw = _pair(k); sh, sw = _pair(s); ph, pw = _pair(p)
        if self._deconv:
            Ho, Wo = (H - 1) * sh - 2 * ph + kh, (W - 1) * sw - 2 * pw + kw
        else:
            Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        out = Slot(ctx.act(N, Ho, Wo, cout))
        return Seq([engine.ConvBN(ctx, "", cin, cout, k, s, p, xin, out, deconv=self._deconv)]), out


class BasicDeconv2d(BasicConv2d):
    """ConvTranspose2d(no bias) -> BatchNorm2d -> ReLU (reference: src/rinet.py:195-205)."""
    _deconv = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        nn.Module.__init__(self)
        self.deconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.activation = nn.ReLU(inplace=True)
        self._geom = (in_channels, out_channels, kernel_size, stride, padding)


def _check_block(conv_block, name):
    if conv_block is not None and conv_block not in (BasicConv2d, BasicDeconv2d):
        raise NotImplementedError(f"{name}: only the reference's BasicConv2d / BasicDeconv2d blocks have HIP plans")


class InceptionA(_PlanModule):
    """reference: src/rinet.py:207-252."""

    def __init__(self, in_channels, identity_downsample=None, conv_block: Optional[Callable[..., nn.Module]] = None):
        super().__init__()
        _check_block(conv_block, "InceptionA")
        c = BasicConv2d
        self.branch1x1 = c(in_channels, 32, kernel_size=1)
        self.branch3x3_1 = c(in_channels, 16, kernel_size=1)
        self.branch3x3_2 = c(16, 32, kernel_size=3, padding=1)
        self.branch3x3dbl_1 = c(in_channels, 32, kernel_size=1)
        self.branch3x3dbl_2 = c(32, 48, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = c(48, 48, kernel_size=3, padding=1)
        self.branch_pool = c(in_channels, 16, kernel_size=1)
        self.identity_downsample = identity_downsample
        self._cin = in_channels

    def _build(self, ctx, xin):
        N, H, W, _ = xin.shape
        out = Slot(ctx.act(N, H, W, 128))
        return engine.inception_a(ctx, "", xin, out, self._cin, self.identity_downsample is not None), out


class InceptionB(_PlanModule):
    """reference: src/rinet.py:254-291.  As there, decoder mode is selected when either conv_block or decoder is
    not None (RIGAN passes its `decoder` flag positionally into conv_block, src/rinet.py:129)."""

    def __init__(self, in_channels, conv_block: Optional[Callable[..., nn.Module]] = None, decoder=None):
        super().__init__()
        self._decoder = not (conv_block is None and decoder is None)
        if self._decoder:
            c, k = BasicDeconv2d, 4
            self.sample = nn.Sequential(nn.Conv2d(in_channels, 128, kernel_size=1, bias=False),
                                        nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True))
        else:
            c, k = BasicConv2d, 3
            self.sample = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.branch3x3_1 = c(in_channels, 32, kernel_size=1)
        self.branch3x3_2 = c(32, 64, kernel_size=k, stride=2, padding=1)
        self.branch3x3dbl_1 = c(in_channels, 32, kernel_size=1)
        self.branch3x3dbl_2 = c(32, 48, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = c(48, 64, kernel_size=k, stride=2, padding=1)
        self._cin = in_channels

    def _build(self, ctx, xin):
        N, H, W, _ = xin.shape
        if self._decoder:
            out = Slot(ctx.act(N, 2 * H, 2 * W, 256))
        else:
            out = Slot(ctx.act(N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 128 + self._cin))
        return engine.inception_b(ctx, "", xin, out, self._cin, self._decoder), out


class InceptionC(_PlanModule):
    """reference: src/rinet.py:293-347."""

    def __init__(self, in_channels, channels_7x7, identity_downsample=None,
                 conv_block: Optional[Callable[..., nn.Module]] = None):
        super().__init__()
        _check_block(conv_block, "InceptionC")
        c, c7 = BasicConv2d, channels_7x7
        self.branch1x1 = c(in_channels, 64, kernel_size=1)
        self.branch7x7_1 = c(in_channels, c7, kernel_size=1)
        self.branch7x7_2 = c(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = c(c7, 64, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = c(in_channels, c7, kernel_size=1)
        self.branch7x7dbl_2 = c(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = c(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = c(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = c(c7, 64, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = c(in_channels, 64, kernel_size=1)
        self.identity_downsample = identity_downsample
        self._cin, self._c7 = in_channels, c7

    def _build(self, ctx, xin):
        N, H, W, _ = xin.shape
        out = Slot(ctx.act(N, H, W, 256))
        return engine.inception_c(ctx, "", xin, out, self._cin, self._c7, self.identity_downsample is not None), out


class InceptionD(_PlanModule):
    """reference: src/rinet.py:349-387 (decoder selection as InceptionB)."""

    def __init__(self, in_channels, conv_block: Optional[Callable[..., nn.Module]] = None, decoder=None):
        super().__init__()
        self._decoder = not (conv_block is None and decoder is None)
        if self._decoder:
            c, k = BasicDeconv2d, 4
            self.sample = nn.Sequential(nn.Conv2d(in_channels, 256, kernel_size=1, bias=False),
                                        nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True))
        else:
            c, k = BasicConv2d, 3
            self.sample = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.branch3x3_1 = c(in_channels, 64, kernel_size=1)
        self.branch3x3_2 = c(64, 128, kernel_size=k, stride=2, padding=1)
        self.branch7x7x3_1 = c(in_channels, 128, kernel_size=1)
        self.branch7x7x3_2 = c(128, 128, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = c(128, 128, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = c(128, 128, kernel_size=k, stride=2, padding=1)
        self._cin = in_channels

    def _build(self, ctx, xin):
        N, H, W, _ = xin.shape
        if self._decoder:
            out = Slot(ctx.act(N, 2 * H, 2 * W, 512))
        else:
            out = Slot(ctx.act(N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 256 + self._cin))
        return engine.inception_d(ctx, "", xin, out, self._cin, self._decoder), out


class InceptionE(_PlanModule):
    """reference: src/rinet.py:389-445."""

    def __init__(self, in_channels, identity_downsample=None, conv_block: Optional[Callable[..., nn.Module]] = None):
        super().__init__()
        _check_block(conv_block, "InceptionE")
        c = BasicConv2d
        self.branch1x1 = c(in_channels, 160, kernel_size=1)
        self.branch3x3_1 = c(in_channels, 192, kernel_size=1)
        self.branch3x3_2a = c(192, 192, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = c(192, 192, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = c(in_channels, 224, kernel_size=1)
        self.branch3x3dbl_2 = c(224, 192, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = c(192, 192, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = c(192, 192, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = c(in_channels, 96, kernel_size=1)
        self.identity_downsample = identity_downsample
        self._cin = in_channels

    def _build(self, ctx, xin):
        N, H, W, _ = xin.shape
        out = Slot(ctx.act(N, H, W, 1024))
        return engine.inception_e(ctx, "", xin, out, self._cin, self.identity_downsample is not None), out


# --------------------------------------------------------------------------- discriminator pieces
class SNConv2d(_PlanModule):
    """Spectral-norm conv + LeakyReLU(0.2) (reference: src/rinet.py:131-148).  HIP plans cover the geometry the
    discriminator uses: 3x3, padding 1, stride 1 or 2, bias, in_channels 1 or a multiple of 8."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        super().__init__()
        if (_pair(kernel_size), _pair(padding), dilation, groups, bias) != ((3, 3), (1, 1), 1, 1, True):
            raise NotImplementedError("SNConv2d: only kernel 3, padding 1, dilation 1, groups 1, bias=True has a HIP plan")
        self.conv2d = nn.utils.spectral_norm(nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding,
                                                       dilation, groups, bias))
        self.activation = nn.LeakyReLU(0.2, inplace=True)
        for m in self.modules():               # re-initialises weight_orig through the aliased .weight, as the reference
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
        self._geom = (in_channels, out_channels, stride if isinstance(stride, int) else stride[0])

    def _make_runner(self, shape, training):
        return _SNRunner(self, shape, training)


class _SNRunner:
    def __init__(self, module, shape, training):
        self.module, self.key_prefix, self.state = module, module._key_prefix, module._state
        cin, cout, stride = module._geom
        N, C, H, W = shape
        dev = self.state.device
        self.ctx = ctx = Ctx(self.state, dev, grads=True, training=training)
        self.first = cin == 1
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        self.a_in = None if self.first else Slot(ctx.act(N, H, W, C))
        self.a_out = Slot(ctx.act(N, Ho, Wo, cout))
        self.layer = engine.SNConv(ctx, ".conv2d", cin, cout, stride, self.a_in, self.a_out, first=self.first)
        ctx.finalize()
        self.N = N
        self.y = torch.empty(N, cout, Ho, Wo, dtype=F32, device=dev)
        self.dx = torch.zeros(N, C, H, W, dtype=F32, device=dev)
        self.serial = 0           # forward calls so far

    def forward(self, x):
        self.serial += 1
        x = x.contiguous().to(F32)
        self.ctx.sn_sigmas()
        self.ctx.begin_forward()
        if self.first:
            self.x = x
            self.layer.forward(x)
        else:
            ops.nchw_to_nhwc(x, self.a_in.v)
            self.layer.forward()
        ops.nhwc_to_nchw(self.a_out.v, self.y)
        return self.y.clone()

    def backward(self, dy, need_dx):
        self.ctx.zero_backward()
        self.state.zero_grad()
        ops.nchw_to_nhwc(dy.to(F32), self.a_out.g)
        if self.first:
            self.dx.zero_()
            self.layer.backward(0, self.N, True, self.dx if need_dx else None)
            self.ctx.sn_finish_wgrads()
            return self.dx.clone() if need_dx else None
        self.layer.backward(0, self.N, True)
        self.ctx.end_backward()
        self.ctx.sn_finish_wgrads()
        if not need_dx:
            return None
        ops.nhwc_to_nchw(self.a_in.g, self.dx)
        return self.dx.clone()


class Self_Attn(_PlanModule):
    """Self attention layer (reference: src/rinet.py:150-181)."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.query_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def _build(self, ctx, xin):
        N, H, W, C = xin.shape
        out = Slot(ctx.act(N, H, W, C))
        layer = engine.Attn(ctx, "", C, xin, out)

        class _P:
            def forward(self_inner):
                layer.forward()

            def backward(self_inner):
                layer.backward(0, N, True)
        return _P(), out


# --------------------------------------------------------------------------- the two networks
class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        ctx.module, ctx.key = module, (tuple(x.shape), module.training)
        ctx.need_dx = x.requires_grad
        y = module._net_forward(x)
        ctx.serial = module._plans()[ctx.key].serial
        return y

    @staticmethod
    def backward(ctx, dy):
        m = ctx.module
        _check_serial(m._plans()[ctx.key], ctx.serial, type(m).__name__)
        dx = m._net_backward(ctx.key, dy.contiguous().to(F32), ctx.need_dx)
        st = m._state
        return (None, dx, *[st.g[k].clone() for k, _ in m.named_parameters()])


class RIGAN(_PlanModule):
    """Residual Inception encoder-decoder generator (reference: src/rinet.py:34-129)."""
    _key_prefix = ""

    def __init__(self, inception_blocks: Optional[List[Callable[..., nn.Module]]] = None):
        super().__init__()
        default = [BasicConv2d, InceptionA, InceptionB, InceptionC, InceptionD, InceptionE, BasicDeconv2d]
        if inception_blocks is None:
            inception_blocks = default
        assert len(inception_blocks) == 7
        if list(inception_blocks) != default:
            raise NotImplementedError("RIGAN: only the reference's default block list has a HIP plan")
        conv_block, a, b, c, d, e, deconv_block = inception_blocks
        self.in_channels, self.inter_channels = 64, 32
        self.conv1 = conv_block(1, 32, 3, 2, padding=1)
        self.conv2 = conv_block(32, 64, 3, 2, padding=1)
        self.layer_block_1 = self._make_layer(a, 1, inter_channels=32)
        self.reduce_grid_1 = self._alter_grid_layer(b, 128, inter_channels=64)
        self.layer_block_2 = self._make_layer(c, 1, inter_channels=64, channels_7x7=64)
        self.reduce_grid_2 = self._alter_grid_layer(d, 256, inter_channels=128)
        self.layer_block_3 = self._make_layer(e, 1, inter_channels=256)
        self.expand_grid_1 = self._alter_grid_layer(d, 1024, inter_channels=128, decoder=1)
        self.layer_block_4 = self._make_layer(c, 1, inter_channels=64, channels_7x7=64)
        self.expand_grid_2 = self._alter_grid_layer(b, 256, inter_channels=64, decoder=1)
        self.layer_block_5 = self._make_layer(a, 1, inter_channels=32)
        self.deconv1 = deconv_block(192, 64, 4, 2, padding=1)
        self.deconv2 = deconv_block(96, 32, 4, 2, padding=1)
        self.out_conv = nn.Conv2d(33, 1, 1, bias=False)

    def _make_layer(self, ri_block, num_residual_blocks, inter_channels, channels_7x7=None):
        """One residual block per stage, with a Conv1x1+BN+ReLU identity path when the width changes
        (reference: src/rinet.py:94-125)."""
        if num_residual_blocks != 1:
            raise NotImplementedError("RIGAN stages hold one residual block each")
        width = inter_channels * 4
        ds = None
        if self.in_channels != width:
            ds = nn.Sequential(nn.Conv2d(self.in_channels, width, 1, 1, bias=False), nn.BatchNorm2d(width), nn.ReLU())
        args = (self.in_channels, channels_7x7, ds) if channels_7x7 is not None else (self.in_channels, ds)
        self.in_channels = width
        return nn.Sequential(ri_block(*args))

    def _alter_grid_layer(self, inception_block, in_channels, inter_channels, decoder=None):
        self.in_channels = inter_channels * 4
        return inception_block(in_channels, decoder)        # decoder lands in conv_block, as in the reference

    def _plans(self):
        if not hasattr(self, "_gplans"):
            object.__setattr__(self, "_gplans", {})
        return self._gplans

    def _net_forward(self, x):
        st = self._ensure_bound()
        key = (tuple(x.shape), self.training)
        plans = self._plans()
        if getattr(self, "_gplans_state", None) is not st:
            plans.clear()
            object.__setattr__(self, "_gplans_state", st)
        G = plans.get(key)
        if G is None:
            N, C, H, W = x.shape
            if C != 1:
                raise ValueError("RIGAN expects one-channel images")
            G = plans[key] = engine.Generator(st, N, H, W, grads=True, training=self.training, device=x.device)
        G.x_keep = x.contiguous().to(F32)
        G.serial = getattr(G, "serial", 0) + 1
        return G.forward(G.x_keep).clone()

    def _net_backward(self, key, dy, need_dx):
        G = self._plans()[key]
        self._state.zero_grad()
        G.backward(dy)
        return None          # the network input is data (conv1 has no data gradient)

    def forward(self, x):
        self._ensure_bound()
        return _NetFn.apply(self, x, *list(self.parameters()))


class ImageDiscriminator(_PlanModule):
    """Spectral-norm discriminator with two self-attention layers (reference: src/rinet.py:10-32)."""
    _key_prefix = ""

    def __init__(self):
        super().__init__()
        self.inter_channels = 64
        c = self.inter_channels
        self.discriminator_net = nn.Sequential(
            SNConv2d(1, c, 3, 1, padding=1),
            SNConv2d(c, c * 2, 3, 2, padding=1),
            SNConv2d(c * 2, c * 4, 3, 2, padding=1),
            Self_Attn(c * 4),
            SNConv2d(c * 4, c * 8, 3, 2, padding=1),
            Self_Attn(c * 8),
            SNConv2d(c * 8, c * 8, 3, 2, padding=1),
        )

    def _plans(self):
        if not hasattr(self, "_dplans"):
            object.__setattr__(self, "_dplans", {})
        return self._dplans

    def _net_forward(self, x):
        st = self._ensure_bound()
        key = (tuple(x.shape), self.training)
        plans = self._plans()
        if getattr(self, "_dplans_state", None) is not st:
            plans.clear()
            object.__setattr__(self, "_dplans_state", st)
        D = plans.get(key)
        if D is None:
            N, C, H, W = x.shape
            D = plans[key] = engine.Discriminator(st, N, H, W, grads=True, training=self.training, device=x.device)
            D.dimg = torch.zeros(N, 1, H, W, dtype=F32, device=x.device)
        D.x_keep = x.contiguous().to(F32)
        D.serial = getattr(D, "serial", 0) + 1
        D.forward(D.x_keep)
        return D.output_nchw()

    def _net_backward(self, key, dy, need_dx):
        D = self._plans()[key]
        v = D.out.act.v
        D.ctx.zero_backward()
        self._state.zero_grad()
        ops.nchw_to_nhwc(dy.reshape(v.N, v.C, v.H, v.W).contiguous(), D.out.g)
        D.dimg.zero_()
        D.backward(0, v.N, wgrad=True, dimg=D.dimg if need_dx else None)
        return D.dimg.clone() if need_dx else None

    def forward(self, x):
        self._ensure_bound()
        return _NetFn.apply(self, x, *list(self.parameters()))


def state_spec(module: nn.Module):
    """[(state_dict key, shape)] of a module (used to seed synthetic weights)."""
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]
