"""Static execution plans for the RIED-GAN hot path on one MI355X.

A plan is built once per (module, batch, H, W): every activation, gradient and
scratch buffer is allocated up front in HBM (NHWC bf16, channel-sliced
concatenation buffers, fp32 master weights in one flat buffer), and forward /
backward are fixed sequences of launches into libried_gan.so on the current
HIP stream with no host synchronisation, so a whole training step can be
captured into one HIP graph.

Layer classes follow the reference's modules (src/rinet.py) one to one:
ConvBN = BasicConv2d / BasicDeconv2d, Inception{A..E}, SNConv, Attn,
Generator = RIGAN, Discriminator = ImageDiscriminator; Vgg is the frozen
VGG16 tap extractor of src/loss.py:69-87.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .ops import View

F32 = torch.float32
_NON_TRAINABLE = ("running_mean", "running_var", "num_batches_tracked", "weight_u", "weight_v")


def is_trainable(name: str) -> bool:
    return not name.endswith(_NON_TRAINABLE)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# --------------------------------------------------------------------------- parameter store
class State:
    """Parameters of one network: trainable tensors are views into one flat fp32 buffer (so Adam and the
    gradient all-reduce are single launches over `flat` / `flat_grad`); buffers keep their own storage."""

    def __init__(self, sd: Dict[str, torch.Tensor], device="cuda"):
        self.device = device
        names = []
        for k in sd:          # q/k/v projection weights (and biases) of an attention layer are laid out back to
            if not is_trainable(k) or k in names:      # back, so the fused [q; k; v] 1x1 conv reads them in place
                continue
            if k.endswith("query_conv.weight"):
                base = k[:-len("query_conv.weight")]
                group = [base + n + s for s in (".weight", ".bias") for n in ("query_conv", "key_conv", "value_conv")]
                assert all(g in sd and g not in names for g in group), group
                names += group
            else:
                names.append(k)
        offs, total = {}, 0
        for k in names:
            offs[k] = total
            total += (sd[k].numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
        self.flat = torch.zeros(total, dtype=F32, device=device)
        self.flat_grad = torch.zeros(total, dtype=F32, device=device)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        self.b: Dict[str, torch.Tensor] = {}
        self.offs = offs
        for k in names:
            n = sd[k].numel()
            self.p[k] = self.flat[offs[k]:offs[k] + n].view(sd[k].shape)
            self.g[k] = self.flat_grad[offs[k]:offs[k] + n].view(sd[k].shape)
            self.p[k].copy_(sd[k].detach().to(F32))
        for k in sd:
            if not is_trainable(k):
                self.b[k] = sd[k].detach().clone().to(device)
        self.adam_m = torch.zeros_like(self.flat)
        self.adam_v = torch.zeros_like(self.flat)
        self.adam_step = torch.zeros(1, dtype=torch.int32, device=device)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {k: v.detach().clone() for k, v in self.p.items()}
        out.update({k: v.detach().clone() for k, v in self.b.items()})
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for k, v in sd.items():
            (self.p if k in self.p else self.b)[k].copy_(v)

    def zero_grad(self):
        self.flat_grad.zero_()

    def adam(self, lr: float, b1=0.01, b2=0.99, eps=1e-8):
        ops.adam(self.flat, self.flat_grad, self.adam_m, self.adam_v, lr, self.adam_step, b1, b2, eps)


# --------------------------------------------------------------------------- buffers
class Act:
    """An activation view and (when gradients are wanted) the gradient view of the same shape."""

    def __init__(self, v: View, g: Optional[View]):
        self.v, self.g = v, g
        # Activation backward fused into the producer of g: the layer whose output this is sets fuse_slope (its
        # ReLU / LeakyReLU slope); the layer that writes g sets `fused` (its stats rows hold the bias gradient's
        # partial sums) and from then on g holds the gradient w.r.t. the PRE-activation values.
        self.fuse_slope: Optional[float] = None
        self.fused: Optional["_Fused"] = None
        # uint8 [N*H*W][C/8], one bit per value of v (positive or not), written by the producer of v: the layer that
        # writes g reads these instead of v for the activation backward (1/16 of the bytes)
        self.signs: Optional[torch.Tensor] = None

    def slice(self, c0, c1) -> "Act":
        return Act(self.v.slice(c0, c1), None if self.g is None else self.g.slice(c0, c1))

    def images(self, n0, n1) -> "Act":
        return Act(self.v.images(n0, n1), None if self.g is None else self.g.images(n0, n1))


class Ctx:
    """Allocation context of one plan.  Gradient views come from one arena that a single memset clears."""

    def __init__(self, state: Optional[State], device="cuda", grads: bool = True, training: bool = True):
        self.st, self.device, self.grads, self.training = state, device, grads, training
        self._grad_views: List[Tuple[View, int]] = []
        self._grad_bytes = 0
        self.grad_arena: Optional[torch.Tensor] = None
        self._f32: List[Tuple[torch.Tensor, int, int]] = []
        self.stats_elems = 0          # fp32 arena of per-tile partial rows (BN sums, bias sums): overwritten
        self.fold_elems = 0           # fp32 arena zeroed at the start of every forward (folded BN sums)
        self.bfold_elems = 0          # fp32 arena zeroed at the start of every backward (folded BN backward sums)
        self.slab_elems = 0
        self.tmp_elems = 0
        self.f32_elems = 0            # shared fp32 scratch (BN backward partial rows, column sums)
        self._stats_req: List[Tuple[object, str, int]] = []
        self.chain = False                 # True: no activation gradient has two writers, nothing needs zeroing
        self.per_image = False             # forward-only plans: BatchNorm statistics per image (see Generator)
        # Synchronised BatchNorm (data parallel): a callable that averages an fp32 tensor over the ranks in place, on
        # the current stream.  Every train-mode layer then runs conv -> bn_sync(its folded sums) -> normalise, and
        # backward reduce -> bn_sync(the folded backward sums) -> apply; the launches divide by the LOCAL pixel count,
        # so rank-averaged sums give exactly the statistics and gradients of the global batch.  (One deviation: the
        # unbiased factor n / (n - 1) of running_var uses the local n.)  Layers run one by one: no sibling groups,
        # no level-wise job tables.
        self.bn_sync = None
        self.dx_group: Optional["InputGradGroup"] = None     # open group of sibling 1x1 layers (see InputGradGroup)
        self.build_lane = 0                # lane (stream + scratch set) the layers being built will run on
        self.packs = ops.JobTable(13)      # every weight repack of the plan: one launch (pack_all)
        self.repack = True                 # False (frozen weights: inference engines): begin_forward() packs nothing
        self.nine: Optional[torch.Tensor] = None      # device scalar 9.0 (divisor of the pool-folded weights)
        self.splitk_target, self.splitk_minpix = 0, 0      # 0: the defaults of _splitk (a launch per layer)
        self.splitk_whole_rounds = False                   # True: split counts that fill whole rounds of the chip (see _splitk)
        self.sn = ops.SnJobs()             # every spectral-norm layer: power iteration / sigma gradient per launch
        # Deferred weight gradients (one launch per tile shape) and the reduction of their split-K slabs (one launch),
        # issued by end_backward().  One pair of tables per gradient bucket: with data parallelism a network's
        # backward pass finishes its gradients bucket by bucket, so each bucket's all-reduce can start while the
        # layers of the next one still run (set_bucket() while building; a single bucket otherwise).
        self._wgrads_all: List[ops.WgradJobs] = []
        self._wreds_all: List[ops.JobTable] = []
        self.set_bucket(0)
        self._late: List = []              # callbacks run by finalize() once every buffer has its address
        self._wred_slab: List[Tuple[ops.JobTable, int, int]] = []   # (table, job, slab offset): patched in finalize()

    def set_bucket(self, b: int):
        """Layers built from here on put their weight gradients into bucket b."""
        while len(self._wgrads_all) <= b:
            self._wgrads_all.append(ops.WgradJobs())
            self._wreds_all.append(ops.JobTable(12))
        self.bucket = b
        self.wgrads, self.wreds = self._wgrads_all[b], self._wreds_all[b]

    @property
    def n_buckets(self) -> int:
        return len(self._wgrads_all)

    # activations
    def act(self, N, H, W, C, grad: Optional[bool] = None) -> Act:
        v = ops.new_view(N, H, W, C, self.device)
        want = self.grads if grad is None else grad
        g = None
        if want:
            g = View(None, self._grad_bytes, C, C, N, H, W)   # ptr = offset until finalize()
            self._grad_views.append((g, self._grad_bytes))
            self._grad_bytes += (N * H * W * C * 2 + 255) // 256 * 256
        return Act(v, g)

    def stats(self, owner, attr, n, zeroed=False):
        """fp32 statistics rows for `owner.attr`; zeroed: from the arena that begin_forward() clears (rows the
        conv epilogues add into), or, zeroed="backward", from the one zero_backward() clears."""
        if zeroed == "backward":
            self._stats_req.append((owner, attr, ("b", self.bfold_elems)))
            self.bfold_elems += (n + 3) // 4 * 4
        elif zeroed:
            self._stats_req.append((owner, attr, -1 - self.fold_elems))
            self.fold_elems += (n + 3) // 4 * 4
        else:
            self._stats_req.append((owner, attr, self.stats_elems))
            self.stats_elems += (n + 3) // 4 * 4
        setattr(owner, attr + "_n", n)

    def alloc_slab(self, n) -> int:
        """Reserve n fp32 of the weight-gradient slab arena; returns the element offset."""
        off = self.slab_elems
        self.slab_elems += (n + 63) // 64 * 64
        return off

    def add_pack(self, src: torch.Tensor, dst: torch.Tensor, N, R, S, C, sn, sr, ss, sc, sigma=None, dst_pitch=None):
        """dst[n][r][s][c] = src[n*sn + r*sr + s*ss + c*sc] / sigma, repacked by pack_all()."""
        self.packs.add([src.data_ptr(), dst.data_ptr(), 0 if sigma is None else sigma.data_ptr(), N, R, S, C,
                        sn, sr, ss, sc, R * S * C if dst_pitch is None else dst_pitch,
                        1 if dst.dtype == F32 else 0], N * R * S * C)

    def add_pack_pair(self, src: torch.Tensor, dst1: Optional[torch.Tensor], dst2: Optional[torch.Tensor], A, B, T,
                      sigma=None):
        """Master [A][B][T] -> dst1 [A][T][B] and dst2 [B][T][A] (bf16, / sigma) from one read of it (a tile job)."""
        assert T <= 16 and A % 2 == 0 and B % 2 == 0
        self.packs.add([src.data_ptr(), 0 if dst1 is None else dst1.data_ptr(), 0 if sigma is None else sigma.data_ptr(),
                        A, B, T, 0 if dst2 is None else dst2.data_ptr(), 0, 0, 0, 0, 0, 2],
                       nblocks=((A + 31) // 32) * ((B + 31) // 32))

    def add_wgrad_reduce(self, slab_off, out: torch.Tensor, splitk, taps, Cp, Cg, st, sp, sg, accumulate=False,
                         split_stride=None, scale=None):
        import struct
        bits = 0 if scale is None else struct.unpack("<i", struct.pack("<f", scale))[0] & 0xFFFFFFFF
        j = self.wreds.add([0, out.data_ptr(), splitk, taps * Cp * Cg if split_stride is None else split_stride, taps,
                            Cp, Cg, st, sp, sg, int(accumulate), bits], nblocks=Cp)
        self._wred_slab.append((self.wreds, j, slab_off))

    def need_tmp(self, n):
        self.tmp_elems = max(self.tmp_elems, n)

    def need_f32(self, n):
        self.f32_elems = max(self.f32_elems, n)

    def finalize(self):
        dev = self.device
        self.grad_arena = torch.zeros(max(self._grad_bytes // 2, 8), dtype=torch.bfloat16, device=dev)
        base = self.grad_arena.data_ptr()
        seen = set()
        for g, off in self._grad_views:
            if id(g) not in seen:
                g.base, g.ptr = self.grad_arena, base + off
                seen.add(id(g))
        self.stats_arena = torch.zeros(max(self.stats_elems, 4), dtype=F32, device=dev)
        self.fold_arena = torch.zeros(max(self.fold_elems, 4), dtype=F32, device=dev)
        self.bfold_arena = torch.zeros(max(self.bfold_elems, 4), dtype=F32, device=dev)
        for owner, attr, off in self._stats_req:
            if isinstance(off, tuple):
                arena, off = self.bfold_arena, off[1]
            else:
                arena, off = (self.stats_arena, off) if off >= 0 else (self.fold_arena, -1 - off)
            setattr(owner, attr, arena[off:off + getattr(owner, attr + "_n")])
        self.slab = torch.empty(max(self.slab_elems, 4), dtype=F32, device=dev)
        for table, j, off in self._wred_slab:
            table.rows[j][0] = self.slab.data_ptr() + 4 * off
        for fn in self._late:
            fn()
        self.packs.finalize(dev)
        self.sn.finalize(dev)
        for t in self._wreds_all + self._wgrads_all:
            t.finalize(dev)
        # one scratch set per lane: branches of a Par run concurrently on their own streams
        self.tmps = [torch.empty(max(self.tmp_elems, 8), dtype=torch.bfloat16, device=dev) for _ in range(N_LANES)]
        self.f32s = [torch.empty(max(self.f32_elems, 4), dtype=F32, device=dev) for _ in range(N_LANES)]
        self.tmp, self.f32 = self.tmps[0], self.f32s[0]

    def tmp_view(self, N, H, W, C, lane: int = 0) -> View:
        t = self.tmps[lane]
        assert N * H * W * C <= t.numel()
        return View(t, t.data_ptr(), C, C, N, H, W)

    def slab_at(self, off) -> int:
        return self.slab.data_ptr() + 4 * off

    def pack_all(self):
        """Rebuild every packed bf16 weight of the plan from the fp32 masters (forward and backward layouts)."""
        ops.pack_jobs(self.packs)

    def begin_forward(self):
        if self.repack:
            self.pack_all()      # per-tile partial-sum rows need no clearing: the conv epilogues overwrite them
        if self.fold_elems and self.training:
            self.fold_arena.zero_()

    def zero_backward(self):
        if not self.chain:       # a chain plan overwrites every gradient buffer before reading it
            self.grad_arena.zero_()
        if self.bfold_elems:     # the BatchNorm backward sums the reductions add into
            self.bfold_arena.zero_()

    def sn_sigmas(self, power_iter: Optional[bool] = None):
        """One power iteration (training; power_iter=False: none) and sigma = u^T W v for every spectral-norm layer;
        must precede pack_all(), which divides the packed weights by sigma."""
        if self.sn.jobs is not None:
            ops.sn_sigma_jobs(self.sn, power_iter=self.training if power_iter is None else bool(power_iter))

    def sn_finish_wgrads(self):
        """After end_backward(): gradients w.r.t. W/sigma (each layer's dwsn) -> gradients of weight_orig."""
        if self.sn.jobs is not None:
            ops.sn_backward_jobs(self.sn)

    def end_backward(self, bucket: Optional[int] = None):
        """The deferred weight gradients (one launch per tile shape), then the reduction of every layer's slabs
        into the master gradients (one launch); of one bucket, or of all."""
        for b in range(self.n_buckets) if bucket is None else (bucket,):
            self._wgrads_all[b].launch()
            ops.wgrad_reduce_jobs(self._wreds_all[b])


N_LANES = 4
# side_stream() numbers beyond the Inception lanes 1..N_LANES-1: whole networks run beside each other
VGG_LANE, DBWD_LANE, REPLICA_LANE0 = N_LANES, N_LANES + 1, N_LANES + 2
# Train-mode BatchNorm layers of at most BN_FOLD_MAX_ELEMS elements fold their statistics into a few rows with
# float atomics in the conv epilogue and skip the finalize launch (about 1 % of the step).  The price: the sums
# depend on the order in which the conv's blocks finish, so two runs differ in the last bits.  BN_FOLD = False
# (or RIED_DETERMINISTIC=1) keeps every reduction in a fixed order: runs are then bit-reproducible.
BN_FOLD = not os.environ.get("RIED_DETERMINISTIC")
# (The bound was 16 M elements; with every layer folded the 512 x 512 inference batch is 1.2 % faster - the 46
# finalize / row-fold launches of its large layers go - and the training step the same.)
BN_FOLD_MAX_ELEMS = int(os.environ.get("RIED_BN_FOLD_MAX_ELEMS", 1 << 30))
PAR_BRANCHES = True      # False: Par runs its branches one after another on the caller's stream (whole plans are
                         # then the unit of concurrency: see train_step.SliceInference)
_side_streams: List["torch.cuda.Stream"] = []


def side_stream(lane: int):
    """Process-wide side streams: 1..N_LANES-1 for the branches of a Par (lane 0 is the caller's current stream);
    higher numbers for whole networks run beside each other (train_step)."""
    while len(_side_streams) < lane:
        _side_streams.append(torch.cuda.Stream())
    return _side_streams[lane - 1]


class _Fused:
    """Stats rows a data-gradient epilogue fills with the column sums of the masked gradient it writes."""

    def __init__(self, ctx: "Ctx", nimg, H, W, Cin, C, k, sh, sw):
        """The data gradient is a transposed conv with Cin input channels (the layer's outputs) producing an
        (nimg, H, W, C) tensor."""
        self.rows = ops.conv2d_stat_rows(nimg, H, W, Cin, C, k, k, sh, sw, True)
        self.C = C
        ctx.stats(self, "stats", self.rows * 2 * C)


# Deferred gradient pointers: Views handed out before Ctx.finalize() hold offsets.  Slices taken before
# finalize() must be re-derived afterwards, so layers keep (parent Act, c0, c1) and slice on use.
class Slot:
    """A channel slice [c0, c1) of an Act, resolved at run time (after the gradient arena exists)."""

    def __init__(self, act: Act, c0: int = 0, c1: Optional[int] = None, n0: int = 0, n1: Optional[int] = None):
        self.act, self.c0, self.c1 = act, c0, act.v.C if c1 is None else c1
        self.n0, self.n1 = n0, act.v.N if n1 is None else n1

    @property
    def C(self):
        return self.c1 - self.c0

    @property
    def v(self) -> View:
        v = self.act.v
        if (self.n0, self.n1) != (0, v.N):
            v = v.images(self.n0, self.n1)
        return v if (self.c0, self.c1) == (0, v.C) else v.slice(self.c0, self.c1)

    @property
    def g(self) -> View:
        g = self.act.g
        if (self.n0, self.n1) != (0, g.N):
            g = g.images(self.n0, self.n1)
        return g if (self.c0, self.c1) == (0, g.C) else g.slice(self.c0, self.c1)

    def sub(self, c0, c1) -> "Slot":
        return Slot(self.act, self.c0 + c0, self.c0 + c1, self.n0, self.n1)

    def images(self, n0, n1) -> "Slot":
        return Slot(self.act, self.c0, self.c1, self.n0 + n0, self.n0 + n1)

    @property
    def shape(self):
        v = self.act.v
        return (self.n1 - self.n0, v.H, v.W, self.C)


def _splitk(blocks: int, kpix: int, target: int = 0, minpix: int = 0, whole_rounds: bool = False) -> int:
    """K splits of a weight gradient whose (tile, tap) grid has `blocks` blocks.  Default: the smallest split count
    that reaches `target` blocks.  whole_rounds (launches that run alone, one per layer: the discriminator's): the
    largest count whose blocks still fit two per CU (512) - measured on the discriminator's four stride-2 layers
    at 256 images, against 384-block launches that leave half the CUs with one block and half with two:
    103 -> 86, 70 -> 62, 64 -> 60 us, the last layer unchanged (432 blocks either way; 576 take 50 us instead of 37)."""
    minpix = minpix or int(os.environ.get("RIED_SPLITK_MINPIX", "768"))       # measured: 256 -> 768 pixels per split, -1 % per step
    if whole_rounds and not target and not os.environ.get("RIED_SPLITK_TARGET"):
        s = max(1, min(512 // max(blocks, 1), (kpix + minpix - 1) // minpix))
        return min(s, 512)
    target = target or int(os.environ.get("RIED_SPLITK_TARGET", "384"))
    s = max(1, min((target + blocks - 1) // blocks, (kpix + minpix - 1) // minpix))
    return min(s, 512)


def _conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


# --------------------------------------------------------------------------- conv (+BN+ReLU) layers
class ConvCore:
    """Weight handling shared by every bf16 conv / deconv layer: packing, forward, dgrad, wgrad."""

    def __init__(self, ctx: Ctx, wname: str, cin, cout, k, stride, pad, deconv: bool, xin: Slot, need_dx: bool,
                 need_wgrad: bool = True, sigma: Optional[torch.Tensor] = None,
                 wgrad_out: Optional[torch.Tensor] = None, pool_fold: bool = False):
        """sigma: device scalar the packed weights are divided by (spectral norm); wgrad_out: where the reduced
        weight gradient goes (master layout; default: the parameter's slot in the state's gradient buffer).
        pool_fold: the layer is avg_pool2d(3, 1, 1) followed by a 1x1 convolution (the branch_pool of Inception A / C /
        E, src/rinet.py:234,250-251 and alike).  With count_include_pad (the default) that is exactly a 3x3
        convolution, zero padding, whose nine taps all hold W / 9: k = 3, pad = 1 here, the master weight stays
        (Cout, Cin, 1, 1); the packed copies replicate it over the taps and the weight gradient sums the taps' slabs
        (times 1/9).  No pooled tensor is written or read, forward or backward."""
        self.ctx, self.wname = ctx, wname
        self.xf_src: Optional["ConvBN"] = None      # set by ConvBN: the layer whose raw output this conv normalises itself
        self.pool_fold = pool_fold
        if pool_fold:
            assert _pair(k) == (3, 3) and _pair(stride) == (1, 1) and _pair(pad) == (1, 1) and not deconv
            if ctx.nine is None:
                ctx.nine = torch.full((1,), 9.0, dtype=F32, device=ctx.device)
            sigma = ctx.nine
        self.cin, self.cout = cin, cout
        self.kh, self.kw = _pair(k)
        self.sh, self.sw = _pair(stride)
        self.ph, self.pw = _pair(pad)
        self.deconv, self.xin, self.need_dx = deconv, xin, need_dx
        N, H, W, C = xin.shape
        assert C == cin, (wname, C, cin)
        self.N, self.H, self.W = N, H, W
        if deconv:
            self.Ho = (H - 1) * self.sh - 2 * self.ph + self.kh
            self.Wo = (W - 1) * self.sw - 2 * self.pw + self.kw
        else:
            self.Ho, self.Wo = _conv_out(H, self.kh, self.sh, self.ph), _conv_out(W, self.kw, self.sw, self.pw)
        taps = self.kh * self.kw
        dev = ctx.device
        self.wp = torch.empty(cout * taps * cin, dtype=torch.bfloat16, device=dev)
        self.wpt = torch.empty(cout * taps * cin, dtype=torch.bfloat16, device=dev) if (need_dx and ctx.grads) else None
        if ctx.grads and need_wgrad:
            # weight gradient: gathered tensor g / pixel-indexed tensor p (see ried_conv2d_wgrad)
            cg, cp = (cout, cin) if deconv else (cin, cout)
            kp = N * (H * W if deconv else self.Ho * self.Wo)
            self.splitk = _splitk(ops.conv2d_wgrad_blocks(cg, cp, self.kh, self.kw, kp), kp, ctx.splitk_target,
                                  ctx.splitk_minpix, whole_rounds=ctx.splitk_whole_rounds)
            self.slab_off = ctx.alloc_slab(self.splitk * taps * cin * cout)
            out = ctx.st.g[wname] if wgrad_out is None else wgrad_out
            if pool_fold: # slab [splitk][9][Cp=cout][Cg=cin]: (split, tap) is one reduction axis -> master (Cout,Cin,1,1)
                ctx.add_wgrad_reduce(self.slab_off, out, self.splitk * taps, 1, cout, cin, 1, cin, 1,
                                     split_stride=cout * cin, scale=1.0 / 9.0)
            elif deconv:  # slab [taps][Cp=cin][Cg=cout] -> master (Cin,Cout,kh,kw)
                ctx.add_wgrad_reduce(self.slab_off, out, self.splitk, taps, cin, cout, 1, cout * taps, taps)
            else:         # slab [taps][Cp=cout][Cg=cin] -> master (Cout,Cin,kh,kw)
                ctx.add_wgrad_reduce(self.slab_off, out, self.splitk, taps, cout, cin, 1, cin * taps, taps)
        if ctx.st is not None:
            w, t, kh, kw = ctx.st.p[wname], taps, self.kh, self.kw
            if pool_fold:   # master (Cout,Cin,1,1) / 9 into every tap of [Cout][3][3][Cin] and of [Cin][3][3][Cout]
                ctx.add_pack(w, self.wp, cout, kh, kw, cin, cin, 0, 0, 1, sigma)
                if self.wpt is not None:
                    ctx.add_pack(w, self.wpt, cin, kh, kw, cout, 1, 0, 0, cin, sigma)
            elif t <= 16 and cin % 2 == 0 and cout % 2 == 0 and PACK_TILES:     # both layouts from one read of the master
                if deconv:
                    ctx.add_pack_pair(w, self.wpt, self.wp, cin, cout, t, sigma)
                else:
                    ctx.add_pack_pair(w, self.wp, self.wpt, cout, cin, t, sigma)
            elif deconv:  # master (Cin,Cout,kh,kw) -> [Cout][kh][kw][Cin]; backward layout [Cin][kh][kw][Cout]
                ctx.add_pack(w, self.wp, cout, kh, kw, cin, t, kw, 1, cout * t, sigma)
                if self.wpt is not None:
                    ctx.add_pack(w, self.wpt, cin, kh, kw, cout, cout * t, kw, 1, t, sigma)
            else:         # master (Cout,Cin,kh,kw) -> [Cout][kh][kw][Cin]; backward layout [Cin][kh][kw][Cout]
                ctx.add_pack(w, self.wp, cout, kh, kw, cin, cin * t, kw, 1, t, sigma)
                if self.wpt is not None:
                    ctx.add_pack(w, self.wpt, cin, kh, kw, cout, t, kw, 1, cin * t, sigma)

    def conv(self, y: View, x: Optional[View] = None, **kw):
        src = self.xf_src
        mix = getattr(self, "xf_mixed", None)
        if mix is not None and x is None:
            # The input is a concatenation: channels [c0, c0 + C) hold the RAW output of `layer` (which stopped at its
            # statistics), the rest finished activations.  One per-channel table - identity for the finished channels
            # (they are >= 0: the ReLU leaves them alone), the layer's BatchNorm for the raw ones - is refreshed from
            # the layer's statistics and parameters, and the kernel normalises its operand on the way in.
            layer, c0, save, gamma, beta = mix
            st, C = self.ctx.st, layer.core.cout
            save.view(2, -1)[:, c0:c0 + C].copy_(layer.save.view(2, C))
            gamma[c0:c0 + C].copy_(st.p[layer.bn + ".weight"])
            beta[c0:c0 + C].copy_(st.p[layer.bn + ".bias"])
            ops.conv2d_xf(self.xin.v, self.wp, y, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw, save, gamma,
                          beta, layer.act, transposed=self.deconv, **kw)
            return
        if src is not None and x is None:    # the producer stopped at its statistics: normalise its raw output on the fly
            st = self.ctx.st
            ops.conv2d_xf(src.raw, self.wp, y, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw, src.save,
                          st.p[src.bn + ".weight"], st.p[src.bn + ".bias"], src.act, transposed=self.deconv, **kw)
            return
        ops.conv2d(self.xin.v if x is None else x, self.wp, y, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw,
                   transposed=self.deconv, **kw)

    def dgrad(self, dy: View, dx: View, accumulate=True, **kw):
        """dx (+)= data gradient of dy (d loss / d conv output); kw: mask / mask_slope / stats / stats_post."""
        ops.conv2d(dy, self.wpt, dx, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw,
                   transposed=not self.deconv, accumulate=accumulate, **kw)

    def defer_wgrad(self, dy_of):
        """Compute this layer's weight gradient in Ctx.end_backward(), together with every other deferred layer,
        instead of in its own launch.  dy_of(): the view that will hold d loss / d conv output at that point (it
        must not be a scratch buffer that later layers overwrite); called once all buffers have addresses."""
        table = self.ctx.wgrads          # the bucket being built now

        def make():
            dy, x = dy_of(), self.xin.v
            g, p = (dy, x) if self.deconv else (x, dy)
            table.add(g, p, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw,
                      self.ctx.slab_at(self.slab_off), self.splitk)
        self.ctx._late.append(make)
        self.deferred = True

    def wgrad(self, dy: View, x: Optional[View] = None):
        """Split-K partial weight gradients into this layer's slab; Ctx.end_backward() reduces all layers."""
        if getattr(self, "deferred", False):
            return
        ctx = self.ctx
        x = self.xin.v if x is None else x
        slab = ctx.slab_at(self.slab_off)
        if self.deconv:
            ops.conv2d_wgrad(dy, x, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw, slab, self.splitk)
        else:
            ops.conv2d_wgrad(x, dy, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw, slab, self.splitk)


class ConvBN:
    """BasicConv2d / BasicDeconv2d (src/rinet.py:183-205): conv(no bias) -> BatchNorm2d -> ReLU.

    The conv epilogue accumulates the per-channel sum / sum of squares; bn_apply writes relu(bn(x)) into `out`
    (a channel slice of the consumer's input buffer), optionally adding onto what is there (residual sum)."""

    def __init__(self, ctx: Ctx, prefix: str, cin, cout, k, stride, pad, xin: Slot, out: Slot, deconv=False,
                 accumulate=False, need_dx=True, conv_key=None, bn_key=None, bias=False, act=1, shared=False,
                 wgrad_out: Optional[torch.Tensor] = None, defer_apply=False, pool_fold=False,
                 xf_src: Optional["ConvBN"] = None, post_pool=False, raw_view: Optional[View] = None):
        """bias / act / shared / wgrad_out serve the legacy networks (src/networks.py:466-542): a conv bias in front
        of the BatchNorm (train mode: it only shifts running_mean), the activation code of ops.bn_apply, layers
        applied several times per forward (BatchNorm and bias gradients then add onto the state's), and a
        private buffer for this use's weight gradient.  defer_apply (forward-only plans): the layer stops at the
        statistics (`save`); its consumer reads `raw` and applies the normalisation itself - xf_src names that
        producer to the consumer, whose convolution then normalises its operand on the way through LDS.
        post_pool (forward-only plans): the layer is the 1x1 convolution of a branch_pool (src/rinet.py:234,250-251 and
        alike) whose avg_pool2d(3, 1, 1) is applied to the convolution's OUTPUT instead of its input - the two are
        linear and commute (no bias; zero padding) - so the pool runs on cout instead of cin channels and the
        convolution can ride in the block's sibling group: conv -> pool + statistics (ops.avgpool3_stats) -> BN.
        raw_view (defer_apply layers): where the raw convolution output goes instead of a buffer of the layer's own -
        the channel slice of a concatenation whose one consumer normalises it on the fly (ConvCore.xf_mixed)."""
        self.ctx, self.prefix, self.lane = ctx, prefix, ctx.build_lane
        conv_key = conv_key or (prefix + (".deconv" if deconv else ".conv"))
        self.bn = bn_key or (prefix + ".bn")
        self.bias_key = conv_key + ".bias" if bias else None
        self.act, self.shared = act, shared
        grp = ctx.dx_group
        grouped = grp is not None and ctx.grads and need_dx and grp.wants(xin, k, stride, pad)
        fwd_member = (grp is not None and not ctx.grads and not bias
                      and grp.wants(xin, k, stride, pad))      # forward-only plans: the group shares the convolution
        self.core = ConvCore(ctx, conv_key + ".weight", cin, cout, k, stride, pad, deconv, xin,
                             need_dx and not grouped, wgrad_out=wgrad_out, pool_fold=pool_fold)
        c = self.core
        c.xf_src = xf_src
        assert xf_src is None or (not ctx.grads and xf_src.defer_apply and xf_src.core.cout == cin)
        assert out.shape == (c.N, c.Ho, c.Wo, cout), (prefix, out.shape, (c.N, c.Ho, c.Wo, cout))
        self.out, self.accumulate = out, accumulate
        assert raw_view is None or (defer_apply and (raw_view.N, raw_view.H, raw_view.W, raw_view.C) == (c.N, c.Ho, c.Wo, cout))
        self.raw = ops.new_view(c.N, c.Ho, c.Wo, cout, ctx.device) if raw_view is None else raw_view
        self.save = torch.empty(2 * cout, dtype=F32, device=ctx.device)
        self.stat_rows = ops.conv2d_stat_rows(c.N, c.Ho, c.Wo, cin, c.cout, c.kh, c.kw, c.sh, c.sw, deconv)
        self.defer_apply = defer_apply
        self.post_pool = bool(post_pool)
        if self.post_pool:
            assert (not ctx.grads and not ctx.per_image and not bias and not defer_apply and not deconv
                    and _pair(k) == (1, 1) and _pair(stride) == (1, 1))
            self.praw = ops.new_view(c.N, c.Ho, c.Wo, cout, ctx.device)
            self.prows = ops.avgpool3_stats_rows(self.praw.npix, cout)
            ctx.stats(self, "pstats", self.prows * 2 * cout)
        self.fused_fwd, self._gstats = None, None      # set by an InputGradGroup that also runs the forward conv
        self.bn_bwd_grouped = False                    # True: the group runs this layer's BatchNorm backward
        self.bn_fwd_grouped = False                    # True: the group runs this layer's BatchNorm forward
        assert not (defer_apply and ctx.grads)
        self.per_image = ctx.per_image and ctx.training
        if self.per_image:     # one (sum, sum of squares) row and one (mean, invstd) pair per image
            assert not ctx.grads and not bias
            ctx.stats(self, "img_stats", c.N * 2 * cout, zeroed=True)
            self.img_save = torch.empty(c.N * 2 * cout, dtype=F32, device=ctx.device)
        self.sliced = (BN_FOLD and ctx.training and self.raw.npix * cout <= BN_FOLD_MAX_ELEMS and not defer_apply
                       and not self.per_image and not self.post_pool)
        if self.sliced:        # every fold row exists: a job-table launch runs 64-pixel tiles, more than the layer's own
            self.stat_rows = ops.bn_fold_rows()
        ctx.stats(self, "stats", self.stat_rows * 2 * cout, zeroed=self.sliced)
        if ctx.bn_sync is not None and ctx.training:
            assert self.sliced and not bias and not shared, "synchronised BatchNorm needs the folded-sums path"
        self.fold = (torch.empty(ops.bn_finalize_fold_floats(cout), dtype=F32, device=ctx.device)
                     if self.stat_rows > 1024 else None)
        self.draw: Optional[View] = None
        if fwd_member:
            grp.join(self)
        self.bwd_fold = self.bwd_one = False
        if ctx.grads:
            ctx.need_f32(ops.bn_backward_scratch_floats(cout))
            if bias:
                ctx.need_f32(ops.colsum_scratch_floats(cout))
            # small layers: the backward sums are folded with float atomics too (one dependent launch fewer)
            self.bwd_fold = BN_FOLD and BN_BWD_FOLD and self.raw.npix * cout <= BN_FOLD_MAX_ELEMS
            assert self.bwd_fold or ctx.bn_sync is None, "synchronised BatchNorm needs the folded backward sums"
            if self.bwd_fold:     # (+ 8 words: the grid barrier's counter of the one-launch form sits behind the rows)
                self.bfold_rows = ops.bn_backward_folded_floats(cout)     # (ctx.stats sets bfold_n = the whole request)
                ctx.stats(self, "bfold", self.bfold_rows + 8, zeroed="backward")
            self.bwd_one = self.bwd_fold and BN_BWD_ONE and ops.bn_backward_fused_ok(self.raw.npix, cout, 1)
            # the gradient w.r.t. the conv output keeps its own buffer, so that the weight gradient can wait for
            # the grouped launch at the end of the backward pass
            if grouped:
                grp.join(self)         # the group hands out a channel slice of its buffer as self.draw when it closes
            else:
                self.draw = ops.new_view(c.N, c.Ho, c.Wo, cout, ctx.device)
            c.defer_wgrad(lambda: self.draw)

    def forward(self):
        ctx, c, st = self.ctx, self.core, self.ctx.st
        if self.per_image:
            if self.fused_fwd is not None:       # raw and the per-image sums are channel slices of the sibling group's
                grp, c0 = self.fused_fwd
                if self._gstats is None:
                    self._gstats = grp.stats[c0:]
                ops.bn_finalize_images(self._gstats, 1, c.N, c.cout, c.Ho * c.Wo, self.img_save, ldstat=grp.total)
            else:
                c.conv(self.raw, stats=self.img_stats, stats_post=3)
                ops.bn_finalize_images(self.img_stats, 1, c.N, c.cout, c.Ho * c.Wo, self.img_save)
            if not self.defer_apply:
                ops.bn_apply_images(self.raw, self.out.v, self.img_save, st.p[self.bn + ".weight"],
                                    st.p[self.bn + ".bias"], relu=self.act, accumulate=self.accumulate)
            return
        if self.post_pool:       # conv (own launch, or a channel slice of the sibling group's) -> pool + statistics -> BN
            if self.fused_fwd is None:
                c.conv(self.raw)
            ops.avgpool3_stats(self.raw, self.praw, self.pstats)
            if ctx.training:
                ops.bn_finalize(self.pstats, self.prows, c.cout, self.praw.npix, self.save,
                                st.b[self.bn + ".running_mean"], st.b[self.bn + ".running_var"],
                                st.b[self.bn + ".num_batches_tracked"])
            else:
                ops.bn_eval_stats(st.b[self.bn + ".running_mean"], st.b[self.bn + ".running_var"], c.cout, self.save)
            ops.bn_apply(self.praw, self.out.v, self.save, st.p[self.bn + ".weight"], st.p[self.bn + ".bias"],
                         relu=self.act, accumulate=self.accumulate)
            return
        # A conv bias in front of a train-mode BatchNorm cancels in the normalised output: the conv runs without
        # it and running_mean, the one place it shows, gets momentum * bias afterwards.
        if self.sliced:
            if self.bn_fwd_grouped:      # normalised with its siblings by InputGradGroup.forward()
                return
            if self.fused_fwd is None:           # (else the sibling group's one convolution has produced raw and the sums)
                c.conv(self.raw, stats=self.stats, stats_post=2)
            if ctx.bn_sync is not None:
                ctx.bn_sync(self.stats)
            self.forward_bn()
            return
        if ctx.training:
            stats, rows, ldstat = self.stats, self.stat_rows, 0
            if self.fused_fwd is not None:       # raw and the partial sums are channel slices of the sibling group's
                grp, c0 = self.fused_fwd
                if self._gstats is None:
                    self._gstats = grp.stats[c0:]
                stats, rows, ldstat = self._gstats, grp.stat_rows, grp.total
            else:
                c.conv(self.raw, stats=self.stats)
            ops.bn_finalize(stats, rows, c.cout, self.raw.npix, self.save, st.b[self.bn + ".running_mean"],
                            st.b[self.bn + ".running_var"], st.b[self.bn + ".num_batches_tracked"], fold=self.fold,
                            ldstat=ldstat)
            if self.bias_key:
                ops.axpy_f32(st.b[self.bn + ".running_mean"], st.p[self.bias_key], 0.1)
        else:
            c.conv(self.raw, bias=st.p[self.bias_key] if self.bias_key else None)
            ops.bn_eval_stats(st.b[self.bn + ".running_mean"], st.b[self.bn + ".running_var"], c.cout, self.save)
        if self.defer_apply:
            return
        ops.bn_apply(self.raw, self.out.v, self.save, st.p[self.bn + ".weight"], st.p[self.bn + ".bias"],
                     relu=self.act, accumulate=self.accumulate)

    def forward_bn(self):
        """Small train-mode layers: statistics from the folded rows, normalisation and activation, in one launch."""
        st = self.ctx.st
        stats, rows, ldstat = self.stats, self.stat_rows, 0
        if self.fused_fwd is not None:
            grp, c0 = self.fused_fwd
            if self._gstats is None:
                self._gstats = grp.stats[c0:]
            stats, rows, ldstat = self._gstats, grp.stat_rows, grp.total
        ops.bn_apply_train(self.raw, self.out.v, stats, rows, self.save,
                           st.p[self.bn + ".weight"], st.p[self.bn + ".bias"], st.b[self.bn + ".running_mean"],
                           st.b[self.bn + ".running_var"], st.b[self.bn + ".num_batches_tracked"], relu=self.act,
                           accumulate=self.accumulate, ldstat=ldstat)
        if self.bias_key:
            ops.axpy_f32(st.b[self.bn + ".running_mean"], st.p[self.bias_key], 0.1)

    def backward_bn(self):
        ctx, st = self.ctx, self.ctx.st
        if ctx.bn_sync is not None:
            gamma, beta, rows = st.p[self.bn + ".weight"], st.p[self.bn + ".bias"], self.bfold[:self.bfold_rows]
            ops.bn_backward_reduce(self.out.g, self.raw, self.save, gamma, beta, rows, relu=self.act)
            ctx.bn_sync(rows)
            ops.bn_backward_apply(self.out.g, self.raw, self.draw, self.save, gamma, beta, rows,
                                  st.g[self.bn + ".weight"], st.g[self.bn + ".bias"], relu=self.act)
            return
        ops.bn_backward(self.out.g, self.raw, self.draw, self.save, st.p[self.bn + ".weight"], st.p[self.bn + ".bias"],
                        ctx.f32s[self.lane], st.g[self.bn + ".weight"], st.g[self.bn + ".bias"], relu=self.act,
                        accumulate=self.shared, folded=self.bfold if self.bwd_fold else None,
                        counter=self.bfold[self.bfold_rows:] if self.bwd_one else None)

    # ---- pieces for Par's level-wise job tables (small train-mode layers only) ----
    def level_ok(self) -> bool:
        return bool(self.ctx.training and self.sliced and not self.per_image and not self.bias_key
                    and not self.defer_apply and not self.shared)

    def fwd_conv_job(self, jobs: "ops.ConvJobs") -> bool:
        """Add this layer's convolution to `jobs`; False when a sibling group's convolution has produced raw."""
        if self.fused_fwd is not None:
            return False
        c = self.core
        jobs.add(c.xin.v, c.wp, self.raw, c.kh, c.kw, c.sh, c.sw, c.ph, c.pw, transposed=c.deconv, stats=self.stats,
                 stats_post=2)
        return True

    def fwd_bn_job(self):
        """The layer's forward BatchNorm as a bn_apply_train_jobs record (None: a sibling group applies it)."""
        if self.bn_fwd_grouped:
            return None
        st, bn = self.ctx.st, self.bn
        stats_ptr, rows, ldstat = self.stats.data_ptr(), self.stat_rows, self.core.cout     # (the table takes no 0 = C)
        if self.fused_fwd is not None:
            grp, c0 = self.fused_fwd
            stats_ptr, rows, ldstat = grp.stats.data_ptr() + 4 * c0, grp.stat_rows, grp.total
        return ops.bn_apply_train_job(self.raw, self.out.v, stats_ptr, rows, ldstat, self.save,
                                      st.b[bn + ".running_mean"], st.b[bn + ".running_var"],
                                      st.b[bn + ".num_batches_tracked"], st.p[bn + ".weight"], st.p[bn + ".bias"],
                                      act=self.act, accumulate=self.accumulate)

    def bwd_bn_job(self):
        """The layer's BatchNorm backward as a bn_backward_jobs record (None: a sibling group runs it)."""
        if self.bn_bwd_grouped:
            return None
        st, bn = self.ctx.st, self.bn
        if not self.bwd_fold and getattr(self, "_bn_scratch", None) is None:
            self._bn_scratch = torch.empty(ops.bn_backward_scratch_floats(self.core.cout), dtype=F32,
                                           device=self.ctx.device)
        # (the scratch word of a folded job carries its barrier counter: read by the one-launch form only)
        return ops.bn_backward_job(self.out.g, self.raw, self.draw, self.save, st.p[bn + ".weight"],
                                   st.p[bn + ".bias"], self.bfold[self.bfold_rows:] if self.bwd_fold else self._bn_scratch,
                                   st.g[bn + ".weight"], st.g[bn + ".bias"], act=self.act, accumulate=self.shared,
                                   folded=self.bfold if self.bwd_fold else None)

    def bwd_dgrad_job(self, jobs: "ops.ConvJobs") -> bool:
        c = self.core
        if not c.need_dx:
            return False
        jobs.add(self.draw, c.wpt, c.xin.g, c.kh, c.kw, c.sh, c.sw, c.ph, c.pw, transposed=not c.deconv,
                 accumulate=True)
        return True

    def backward(self):
        """Reads out.g; writes weight / BN gradients; accumulates into xin.g when need_dx."""
        self.backward_local()
        self.backward_input()

    def backward_local(self):
        """Everything that touches only this layer's own buffers (safe to run beside sibling branches)."""
        ctx, c, st = self.ctx, self.core, self.ctx.st
        draw = self.draw
        if self.bn_bwd_grouped:      # done for all siblings at once by InputGradGroup.backward()
            return
        self.backward_bn()
        if self.bias_key:        # zero up to rounding in train mode, as in torch's autograd
            ops.colsum(draw, st.g[self.bias_key], ctx.f32s[self.lane])
        c.wgrad(draw)

    def backward_input(self):
        """xin.g += data gradient; the lane's temp still holds the pre-BN gradient from backward_local()."""
        c = self.core
        if c.need_dx:
            c.dgrad(self.draw, c.xin.g)


class InputGradGroup:
    """Sibling 1x1 convolutions that read the same tensor - the first layers of an Inception block's branches and its
    identity_downsample (src/rinet.py:99-103, 225-233).  Their gradients w.r.t. that tensor are ONE transposed 1x1
    convolution over the concatenation of their pre-BatchNorm gradients (which they write into channel slices of
    one buffer) with their weights concatenated along K, instead of one accumulating launch - a read-modify-write
    of the whole input gradient - per sibling.
    Forward, for small train-mode layers (the folded-statistics path): the siblings' convolutions are one launch too
    - weights concatenated along the output channels, each sibling's pre-BatchNorm tensor and statistics a channel
    slice of the group's - and the input is read once instead of once per sibling."""

    def __init__(self, ctx: Ctx, xin: Slot):
        self.ctx, self.xin, self.members = ctx, xin, []
        self.drawcat: Optional[View] = None
        self.fwd = False

    def wants(self, xin: Slot, k, stride, pad) -> bool:
        a, b = self.xin, xin
        return (a.act is b.act and (a.c0, a.c1, a.n0, a.n1) == (b.c0, b.c1, b.n0, b.n1) and
                _pair(k) == (1, 1) and _pair(stride) == (1, 1) and _pair(pad) == (0, 0))

    def join(self, layer: "ConvBN"):
        self.members.append(layer)

    def close(self):
        """Allocate the shared gradient buffer, hand out its slices, register the packs of the concatenated
        backward weights; returns the layer to put FIRST in the block's sequence (it runs last in backward)."""
        ctx = self.ctx
        ctx.dx_group = None
        N, H, W, cin = self.xin.shape
        total = sum(m.core.cout for m in self.members)
        c0 = 0
        if ctx.grads:
            assert len(self.members) >= 1
            self.drawcat = ops.new_view(N, H, W, total, ctx.device)
            self.wcat = torch.empty(cin * total, dtype=torch.bfloat16, device=ctx.device)     # [cin][total]
        for m in self.members if ctx.grads else ():
            co = m.core.cout
            m.draw = self.drawcat.slice(c0, c0 + co)
            if ctx.st is not None:      # master (Cout, Cin, 1, 1) -> column slice [c0, c0 + Cout) of row c_in
                rs, cs = (co, 1) if m.core.deconv else (1, cin)     # (a 1x1 transposed conv's master is (Cin, Cout, 1, 1))
                ctx.add_pack(ctx.st.p[m.core.wname], self.wcat[c0:], cin, 1, 1, co, rs, 1, 1, cs, dst_pitch=total)
            c0 += co
        self.fwd = (ctx.st is not None and ctx.training and len(self.members) >= 2 and
                    not any(m.bias_key for m in self.members))
        self.images = self.fwd and ctx.per_image          # one statistics row per image
        self.folded = self.fwd and not self.images and all(m.sliced or m.post_pool for m in self.members)
        self.fwd_jobs = None
        if self.fwd:
            self.total = total
            self.rawcat = ops.new_view(N, H, W, total, ctx.device)
            self.wfwd = torch.empty(total * cin, dtype=torch.bfloat16, device=ctx.device)           # [total][cin]
            self.stat_rows = ops.conv2d_stat_rows(N, H, W, cin, total, 1, 1)
            if self.images:
                ctx.stats(self, "stats", N * 2 * total, zeroed=True)
            elif self.folded:  # small layers: atomically folded rows, finished inside the siblings' apply kernels
                self.stat_rows = min(self.stat_rows, ops.bn_fold_rows())
                ctx.stats(self, "stats", ops.bn_fold_rows() * 2 * total, zeroed=True)
            else:              # one row per tile; every sibling finalizes its channel slice of them
                ctx.stats(self, "stats", self.stat_rows * 2 * total)
                for m in self.members:
                    m.sliced = False
                    if self.stat_rows > 1024 and m.fold is None:
                        m.fold = torch.empty(ops.bn_finalize_fold_floats(m.core.cout), dtype=F32, device=ctx.device)
            c0 = 0
            for m in self.members:
                co = m.core.cout
                m.raw = self.rawcat.slice(c0, c0 + co)
                m.fused_fwd = (self, c0)
                rs, cs = (1, co) if m.core.deconv else (cin, 1)
                ctx.add_pack(ctx.st.p[m.core.wname], self.wfwd[c0 * cin:], co, 1, 1, cin, rs, 1, 1, cs)
                c0 += co
            # the BatchNorm forward of the siblings that do not add onto the block's output: one job table
            if self.folded and FWD_BN_JOBS and sum(not m.accumulate for m in self.members) >= 2:
                for m in self.members:
                    m.bn_fwd_grouped = not m.accumulate
                ctx._late.append(self._make_fwd_jobs)
        # the siblings' BatchNorm backward passes as one job table (three launches instead of three per sibling)
        self.bn_jobs = None
        if ctx.grads and ctx.st is not None and len(self.members) >= 2 and not any(m.bias_key for m in self.members):
            for m in self.members:
                m.bn_bwd_grouped = True
                m._bn_scratch = (None if m.bwd_fold else
                                 torch.empty(ops.bn_backward_scratch_floats(m.core.cout), dtype=F32, device=ctx.device))
            ctx._late.append(self._make_bn_jobs)        # the gradient views have addresses only after finalize()
        return self

    def _make_bn_jobs(self):
        st = self.ctx.st
        rows = [ops.bn_backward_job(m.out.g, m.raw, m.draw, m.save, st.p[m.bn + ".weight"], st.p[m.bn + ".bias"],
                                    m.bfold[m.bfold_rows:] if m.bwd_fold else m._bn_scratch, st.g[m.bn + ".weight"],
                                    st.g[m.bn + ".bias"], act=m.act, accumulate=m.shared,
                                    folded=m.bfold if m.bwd_fold else None)
                for m in self.members]
        self.bn_jobs = torch.tensor(rows, dtype=torch.int64).to(self.ctx.device)
        self.bn_maxc = max(m.core.cout for m in self.members)
        self.npix = self.drawcat.npix
        self.bn_all_folded = bn_jobs_mode(self.members, self.npix, self.bn_maxc)

    def _make_fwd_jobs(self):
        st, rows, c0, cs = self.ctx.st, [], 0, []
        for m in self.members:
            if m.bn_fwd_grouped:
                rows.append(ops.bn_apply_train_job(
                    m.raw, m.out.v, self.stats.data_ptr() + 4 * c0, self.stat_rows, self.total, m.save,
                    st.b[m.bn + ".running_mean"], st.b[m.bn + ".running_var"], st.b[m.bn + ".num_batches_tracked"],
                    st.p[m.bn + ".weight"], st.p[m.bn + ".bias"], act=m.act))
                cs.append(m.core.cout)
            c0 += m.core.cout
        self.fwd_jobs = torch.tensor(rows, dtype=torch.int64).to(self.ctx.device)
        self.fwd_c = (len(rows), max(cs), min(cs))

    def forward(self):
        if self.fwd:
            ops.conv2d(self.xin.v, self.wfwd, self.rawcat, 1, 1, 1, 1, 0, 0, stats=self.stats,
                       stats_post=3 if self.images else 2 if self.folded else 0)
            if self.fwd_jobs is not None:
                ops.bn_apply_train_jobs(self.fwd_jobs, self.fwd_c[0], self.rawcat.npix, self.fwd_c[1], self.fwd_c[2])

    def backward(self):
        if not self.ctx.grads:
            return
        if self.bn_jobs is not None:
            ops.bn_backward_jobs(self.bn_jobs, len(self.members), self.npix, self.bn_maxc, self.bn_all_folded)
        ops.conv2d(self.drawcat, self.wcat, self.xin.g, 1, 1, 1, 1, 0, 0, transposed=True, accumulate=True)


def bn_jobs_mode(layers, npix: int, maxc: int) -> int:
    """The all_folded argument of ops.bn_backward_jobs for these layers: 0 per-block rows, 1 folded rows (two
    launches), 2 folded rows and one launch with a grid barrier (layers small enough for it)."""
    if not all(l.bwd_fold for l in layers):
        return 0
    return 2 if BN_BWD_ONE and ops.bn_backward_fused_ok(npix, maxc, len(layers)) else 1


if os.environ.get("RIED_NO_WGRAD_ROWS"):      # A/B switch: every weight gradient one tap per block
    ops.lib.ried_conv2d_wgrad_row_blocks(0)
SIBLING_GROUPS = not os.environ.get("RIED_NO_SIBLING_GROUPS")      # switch for A/B runs and tests
FUSED_ATTN = not os.environ.get("RIED_NO_FUSED_ATTN")
PACK_TILES = not os.environ.get("RIED_NO_PACK_TILES")
FWD_BN_JOBS = not os.environ.get("RIED_NO_FWD_BN_JOBS")
BN_BWD_FOLD = not os.environ.get("RIED_NO_BN_BWD_FOLD")
LEVEL_JOBS = not os.environ.get("RIED_NO_LEVEL_JOBS")
POOL_FOLD = not os.environ.get("RIED_NO_POOL_FOLD")
POOL_AFTER = not os.environ.get("RIED_NO_POOL_AFTER")      # forward-only plans: branch_pool as conv -> pool (A/B switch)
# BatchNorm backward of small layers as one launch (partial sums, grid barrier, apply) instead of two.  Off: measured
# 7.81 against 7.79 ms per step - the barrier's atomic round trips cost what the launch boundary they replace costs
# (43 launches fewer per step; RIED_BN_BWD_ONE=1 turns it on).
BN_BWD_ONE = bool(os.environ.get("RIED_BN_BWD_ONE"))


def open_input_grad_group(ctx: Ctx, xin: Slot) -> Optional[InputGradGroup]:
    if not SIBLING_GROUPS or ctx.bn_sync is not None or not (ctx.grads or (ctx.training and ctx.st is not None)):
        return None
    ctx.dx_group = InputGradGroup(ctx, xin)
    return ctx.dx_group


def close_input_grad_group(grp: Optional[InputGradGroup], seq: "Seq"):
    if grp is not None:
        grp.close()
        if grp.ctx.grads or grp.fwd:
            seq.layers.insert(0, grp)


class ConvBNC1:
    """Generator conv1: 3x3 s2 p1 on the one-channel fp32 image, then BN + ReLU (src/rinet.py:55)."""

    def __init__(self, ctx: Ctx, prefix: str, cout, N, H, W, out: Slot):
        self.ctx, self.prefix, self.cout = ctx, prefix, cout
        self.N, self.H, self.W = N, H, W
        self.Ho, self.Wo = _conv_out(H, 3, 2, 1), _conv_out(W, 3, 2, 1)
        assert out.shape == (N, self.Ho, self.Wo, cout)
        self.out = out
        self.raw = ops.new_view(N, self.Ho, self.Wo, cout, ctx.device)
        self.save = torch.empty(2 * cout, dtype=F32, device=ctx.device)
        self.stat_rows = ops.conv_c1_stat_rows(N, H, W, cout, 2)
        ctx.stats(self, "stats", self.stat_rows * 2 * cout)
        self.per_image = ctx.per_image and ctx.training
        if self.per_image:     # the kernel writes one row per 256 consecutive pixels: whole rows per image
            assert self.stat_rows % N == 0 and (self.Ho * self.Wo) % 256 == 0, "per-image statistics: map too small"
            self.img_save = torch.empty(N * 2 * cout, dtype=F32, device=ctx.device)
        self.fold = (torch.empty(ops.bn_finalize_fold_floats(cout), dtype=F32, device=ctx.device)
                     if self.stat_rows > 1024 else None)
        if ctx.grads:
            ctx.need_f32(ops.bn_backward_scratch_floats(cout))
            ctx.need_f32(ops.conv_c1_wgrad_scratch_floats(cout))
            ctx.need_tmp(N * self.Ho * self.Wo * cout)
            if ctx.bn_sync is not None:
                ctx.stats(self, "bfold", ops.bn_backward_folded_floats(cout), zeroed="backward")
        self.x: Optional[torch.Tensor] = None

    def forward(self, x: torch.Tensor):
        ctx, st, bn = self.ctx, self.ctx.st, self.prefix + ".bn"
        self.x = x
        w = st.p[self.prefix + ".conv.weight"]
        if self.per_image:
            ops.conv_c1_fwd(x, w, self.raw, 2, stats=self.stats)
            ops.bn_finalize_images(self.stats, self.stat_rows // self.N, self.N, self.cout, self.Ho * self.Wo,
                                   self.img_save)
            ops.bn_apply_images(self.raw, self.out.v, self.img_save, st.p[bn + ".weight"], st.p[bn + ".bias"], relu=True)
            return
        if ctx.training:
            ops.conv_c1_fwd(x, w, self.raw, 2, stats=self.stats)
            if ctx.bn_sync is not None:
                ctx.bn_sync(self.stats)
            ops.bn_finalize(self.stats, self.stat_rows, self.cout, self.raw.npix, self.save, st.b[bn + ".running_mean"],
                            st.b[bn + ".running_var"], st.b[bn + ".num_batches_tracked"], fold=self.fold)
        else:
            ops.conv_c1_fwd(x, w, self.raw, 2)
            ops.bn_eval_stats(st.b[bn + ".running_mean"], st.b[bn + ".running_var"], self.cout, self.save)
        ops.bn_apply(self.raw, self.out.v, self.save, st.p[bn + ".weight"], st.p[bn + ".bias"], relu=True)

    def backward(self):
        ctx, st, bn = self.ctx, self.ctx.st, self.prefix + ".bn"
        draw = ctx.tmp_view(self.N, self.Ho, self.Wo, self.cout)
        if ctx.bn_sync is not None:
            ops.bn_backward_reduce(self.out.g, self.raw, self.save, st.p[bn + ".weight"], st.p[bn + ".bias"], self.bfold)
            ctx.bn_sync(self.bfold)
            ops.bn_backward_apply(self.out.g, self.raw, draw, self.save, st.p[bn + ".weight"], st.p[bn + ".bias"],
                                  self.bfold, st.g[bn + ".weight"], st.g[bn + ".bias"])
        else:
            ops.bn_backward(self.out.g, self.raw, draw, self.save, st.p[bn + ".weight"], st.p[bn + ".bias"], ctx.f32,
                            st.g[bn + ".weight"], st.g[bn + ".bias"], relu=True)
        ops.conv_c1_wgrad(self.x, draw, 2, st.g[self.prefix + ".conv.weight"], ctx.f32)   # grads were zeroed


class ConvPlain:
    """A bare 1x1 Conv2d without bias, BN or activation (the `sample.0` of decoder blocks, src/rinet.py:265,360)."""

    def __init__(self, ctx: Ctx, wname: str, cin, cout, xin: Slot, out: Slot):
        self.ctx, self.out = ctx, out
        self.core = ConvCore(ctx, wname, cin, cout, 1, 1, 0, False, xin, True)
        if ctx.grads:
            self.core.defer_wgrad(lambda: self.out.g)     # the gradient arena keeps out.g until the next backward

    def forward(self):
        self.core.conv(self.out.v)

    def backward(self):
        self.backward_local()
        self.backward_input()

    def backward_local(self):
        self.core.wgrad(self.out.g)

    def backward_input(self):
        self.core.dgrad(self.out.g, self.core.xin.g)


class Pool:
    def __init__(self, ctx: Ctx, mode: int, xin: Slot, out: Slot):
        self.mode, self.xin, self.out = mode, xin, out

    def forward(self):
        ops.pool_fwd(self.xin.v, self.out.v, self.mode)

    def backward(self):
        ops.pool_bwd(None if self.mode == ops.POOL_AVG3 else self.xin.v, self.out.g, self.xin.g, self.mode)

    def backward_local(self):
        pass

    backward_input = backward


class Upsample:
    def __init__(self, ctx: Ctx, xin: Slot, out: Slot):
        self.xin, self.out = xin, out

    def forward(self):
        ops.upsample2x_fwd(self.xin.v, self.out.v)

    def backward(self):
        ops.upsample2x_bwd(self.out.g, self.xin.g)


class Seq:
    """Layers run in order forward and in reverse order backward."""

    def __init__(self, layers=None):
        self.layers = list(layers or [])

    def add(self, layer):
        self.layers.append(layer)
        return layer

    def forward(self):
        for l in self.layers:
            l.forward()

    def backward(self):
        for l in reversed(self.layers):
            l.backward()

    def backward_local(self):
        """All but the write into this sequence's input gradient (which sibling branches also write)."""
        for l in reversed(self.layers[1:]):
            l.backward()
        self.layers[0].backward_local()

    def backward_input(self):
        self.layers[0].backward_input()


class Par:
    """Branches that read one tensor and write disjoint channel slices of another (the Inception branches,
    src/rinet.py:225-252 and alike).  Each branch runs on its own stream, with its own scratch lane; the streams
    fork from and rejoin the caller's stream, so a graph capture records the branches as parallel paths.  The
    branches' gradients w.r.t. the shared input are added one after another once they have joined."""

    def __init__(self, ctx: Ctx, branches: List[Seq]):
        assert 1 <= len(branches) <= N_LANES
        self.ctx, self.branches = ctx, branches
        self.levels = None
        if LEVEL_JOBS and ctx.grads and ctx.training and ctx.st is not None and not ctx.per_image \
                and ctx.bn_sync is None:
            ctx._late.append(self._build_levels)

    # Level-wise execution (small train-mode layers).  The k-th layers of the branches do not depend on each other,
    # and on small maps each is a short, latency-bound launch: their convolutions run as ONE job-table grid, their
    # BatchNorms as another, forward and backward, on the caller's stream.  Layers the job tables do not take (pools,
    # upsampling, large layers) run on their own in the same level order.
    def _build_levels(self):
        dev = self.ctx.device
        depth = max(len(b.layers) for b in self.branches)
        levels, worth, carry = [], False, []
        for k in range(depth):
            items = [b.layers[k] for b in self.branches if k < len(b.layers)]
            ok = [l for l in items if isinstance(l, ConvBN) and l.level_ok()]
            rest = [l for l in items if not (isinstance(l, ConvBN) and l.level_ok())]
            lv = {"rest": rest, "ok": ok, "k": k}
            # forward: convolutions, then BatchNorms grouped by pixel count
            fc = ops.ConvJobs()
            lv["fwd_conv_single"] = []
            convs = [l for l in ok if l.fused_fwd is None]
            if len(convs) >= 2:
                for l in convs:
                    l.fwd_conv_job(fc)
                fc.finalize(dev)
                lv["fwd_conv"] = fc
            else:
                lv["fwd_conv"] = None
                lv["fwd_conv_single"] = convs
            # (the first layers' BatchNorms that add onto the block output - after the identity path has written it,
            # which happened before this Par - ride in the next level's table instead of a launch of their own)
            pairs = carry + [(l, l.fwd_bn_job()) for l in ok if not l.bn_fwd_grouped]
            carry = []
            if k == 0 and depth > 1 and pairs and all(l.accumulate for l, _ in pairs):
                carry, pairs = pairs, []
            lv["fwd_bn"], lv["fwd_bn_single"] = self._bn_tables(pairs, dev)
            # backward: BatchNorms, then data gradients (level 0: the branches' inputs are one tensor - one by one)
            lv["bwd_bn"], lv["bwd_bn_single"] = self._bn_tables(
                [(l, l.bwd_bn_job()) for l in ok if not l.bn_bwd_grouped], dev)
            lv["bwd_all_folded"] = {n: bn_jobs_mode([l for l in ok if not l.bn_bwd_grouped and l.raw.npix == n], n, t[2])
                                    for n, t in lv["bwd_bn"].items()}
            bc = ops.ConvJobs()
            dgs = [l for l in ok if l.core.need_dx] if k > 0 else []
            if len(dgs) >= 2:
                for l in dgs:
                    l.bwd_dgrad_job(bc)
                bc.finalize(dev)
                lv["bwd_conv"], lv["bwd_conv_single"] = bc, []
            else:
                lv["bwd_conv"], lv["bwd_conv_single"] = None, dgs
            worth = worth or lv["fwd_conv"] is not None or bool(lv["fwd_bn"]) or bool(lv["bwd_bn"]) \
                or lv["bwd_conv"] is not None
            levels.append(lv)
        if worth:
            self.levels = levels

    @staticmethod
    def _bn_tables(pairs, dev):
        """[(layer, record)] -> ({npix: (device table, njobs, maxC, minC)} for counts >= 2, [layers left alone])."""
        by = {}
        for l, row in pairs:
            by.setdefault(l.raw.npix, []).append((l, row))
        tables, singles = {}, []
        for npix, lr in by.items():
            if len(lr) >= 2:
                cs = [l.core.cout for l, _ in lr]
                tables[npix] = (torch.tensor([r for _, r in lr], dtype=torch.int64).to(dev), len(lr), max(cs), min(cs))
            else:
                singles += [l for l, _ in lr]
        return tables, singles

    def _forward_levels(self):
        for lv in self.levels:
            for l in lv["rest"]:
                l.forward()
            if lv["fwd_conv"] is not None:
                lv["fwd_conv"].launch()
            for l in lv["fwd_conv_single"]:
                l.core.conv(l.raw, stats=l.stats, stats_post=2)
            for npix, (table, n, maxc, minc) in lv["fwd_bn"].items():
                ops.bn_apply_train_jobs(table, n, npix, maxc, minc)
            for l in lv["fwd_bn_single"]:
                l.forward_bn()

    def _backward_levels(self):
        for lv in reversed(self.levels):
            first = lv["k"] == 0
            for l in lv["rest"]:
                if first:
                    l.backward_local()
                else:
                    l.backward()
            for npix, (table, n, maxc, minc) in lv["bwd_bn"].items():
                ops.bn_backward_jobs(table, n, npix, maxc, lv["bwd_all_folded"][npix])
            for l in lv["bwd_bn_single"]:
                l.backward_bn()
            for l in lv["ok"]:
                l.core.wgrad(l.draw)           # (deferred to the grouped launch in every plan built here: a no-op)
            if lv["bwd_conv"] is not None:
                lv["bwd_conv"].launch()
            for l in lv["bwd_conv_single"]:
                l.backward_input()
        for b in self.branches:
            b.backward_input()

    def _run(self, fn_name: str):
        if os.environ.get("RIED_SERIAL_BRANCHES") or not PAR_BRANCHES:     # one stream (debugging aid; replica lanes)
            for b in self.branches:
                getattr(b, fn_name)()
            return
        main = torch.cuda.current_stream()
        for lane in range(1, len(self.branches)):
            s = side_stream(lane)
            s.wait_stream(main)
            with torch.cuda.stream(s):
                getattr(self.branches[lane], fn_name)()
        getattr(self.branches[0], fn_name)()
        for lane in range(1, len(self.branches)):
            main.wait_stream(side_stream(lane))

    def forward(self):
        if self.levels is not None:
            return self._forward_levels()
        self._run("forward")

    def backward(self):
        if self.levels is not None:
            return self._backward_levels()
        self._run("backward_local")
        for b in self.branches:
            b.backward_input()


class _Lane:
    """with _Lane(ctx, i): layers built inside run on lane i (stream and scratch set)."""

    def __init__(self, ctx: Ctx, lane: int):
        self.ctx, self.lane = ctx, lane

    def __enter__(self):
        self.ctx.build_lane = self.lane
        return Seq()

    def __exit__(self, *a):
        self.ctx.build_lane = 0


class Identity:
    """Residual identity without a downsample conv: out = x (then branches add onto it)."""

    def __init__(self, xin: Slot, out: Slot):
        self.xin, self.out = xin, out

    def forward(self):
        ops.ew(ops.EW_COPY, self.xin.v, self.out.v)

    def backward(self):
        ops.ew(ops.EW_ADD, self.out.g, self.xin.g)


def _chain(ctx: Ctx, seq: Seq, prefix: str, specs, xin: Slot, out: Slot, deconv=False, accumulate=False):
    """specs: [(name, cin, cout, k, stride, pad)]; the last layer writes `out`, inner layers get own buffers."""
    cur, prev = xin, None
    for i, (name, cin, cout, k, s, p) in enumerate(specs):
        last = i == len(specs) - 1
        N, H, W, _ = cur.shape
        kh, kw = _pair(k); sh, sw = _pair(s); ph, pw = _pair(p)
        if deconv:
            Ho, Wo = (H - 1) * sh - 2 * ph + kh, (W - 1) * sw - 2 * pw + kw
        else:
            Ho, Wo = _conv_out(H, kh, sh, ph), _conv_out(W, kw, sw, pw)
        dst = out if last else Slot(ctx.act(N, Ho, Wo, cout))
        # forward-only plans: an inner layer whose one consumer runs the general convolution kernel stops at its
        # statistics, and the consumer normalises the raw output itself (no second pass over the tensor)
        defer = not last and _xf_consumer_ok(ctx, specs[i + 1], deconv, Ho, Wo)
        prev = seq.add(ConvBN(ctx, prefix + "." + name, cin, cout, k, s, p, cur, dst, deconv=deconv,
                              accumulate=accumulate and last, defer_apply=defer,
                              xf_src=prev if (prev is not None and prev.defer_apply) else None))
        cur = dst
    return cur


# The consumer of an inner layer can apply that layer's BatchNorm + ReLU to its operand on the way in (ops.conv2d_xf:
# global -> registers -> scale, shift, ReLU -> LDS), so that the normalised tensor is never written or read.  The
# general convolution kernel fetches its pixel operand once per tap, so the transform is repeated per tap as well
# and the kernel becomes vector-arithmetic bound: measured alone (tools/xf_bench.py, 32 slices of 512 x 512), a
# stride-1 3 x 3 or 1 x 7 consumer takes 1.5-2.2x as long as the plain kernel (52 -> 117 us where the apply pass it
# saves takes 14 us), a stride-2 3 x 3 consumer - 2.25 taps per operand pixel - breaks even or wins (45 -> 41 us,
# 22 -> 19 us for apply + convolution).  Whole batch: 5.13 ms without, 5.105 ms with the stride-2 consumers only,
# 5.74 ms with every consumer.  So: stride-2 consumers by default (RIED_XF_FUSE=0: none, =1: all).  An earlier
# form that transformed the operand in place in LDS cost the consumers 40-45 us each (LDS bandwidth).  Results are
# bit-identical to apply + convolution (same fp32 scale / shift, same rounding to bf16).
# The discriminator's first layer also writes one sign bit per activation and the second layer's data gradient reads
# those for the LeakyReLU backward instead of the 64-channel activation itself (RIED_SIGN_MASK=0: the activation).
SIGN_MASK = os.environ.get("RIED_SIGN_MASK", "1") != "0"
# A discriminator backward pass that wants parameter gradients only forms the first layer's weight gradient inside the
# second layer's data-gradient kernel and never stores the gradient of the first activation (RIED_C1_FOLD=0: off).
C1_FOLD = os.environ.get("RIED_C1_FOLD", "1") != "0"
# Forward-only plans: deconv2 normalises deconv1's raw output inside the deconvolution halo kernel (RIED_NO_D4_XF=1: off).
D4_XF = not os.environ.get("RIED_NO_D4_XF")
XF_FUSE_MODE = os.environ.get("RIED_XF_FUSE", "s2")
XF_FUSE = XF_FUSE_MODE not in ("", "0")


def _xf_consumer_ok(ctx: Ctx, spec, deconv: bool, H: int, W: int) -> bool:
    """Will the layer `spec` (input H x W) take the general convolution kernel - the one that can normalise its
    operand on the fly (ops.conv2d_xf)?  Mirrors the dispatch of ried_conv2d: 3x3 stride-1 layers on maps of whole
    tiles and ConvTranspose2d(4, 2, 1) on 16 x 16 tiles have kernels of their own."""
    if not XF_FUSE or ctx.grads or not ctx.training or ctx.per_image or ctx.st is None:
        return False
    _, cin, cout, k, s, p = spec
    (kh, kw), (sh, sw) = _pair(k), _pair(s)
    if cin > 1024:
        return False
    if XF_FUSE_MODE == "s2" and (deconv or sh != 2):      # stride-2 consumers only: each operand pixel serves 2.25 taps
        return False
    if not deconv and (kh, kw, sh, sw) == (3, 3, 1, 1) and cin % 64 == 0 and H % 8 == 0 and W % 8 == 0:
        return False
    if deconv and (kh, kw, sh, sw) == (4, 4, 2, 2) and cin % 32 == 0 and H % 16 == 0 and W % 16 == 0:
        return False
    return True


def _pool_branch(ctx, seq, prefix, xin: Slot, out_slice: Slot, cin, cout):
    """branch_pool = avg_pool2d(3, 1, 1) -> BasicConv2d 1x1 (src/rinet.py:234,250-251; 320,344-345; 416,441-442), added
    onto the block output.  Forward-only plans fold it into one 3x3 convolution where the halo-tile kernel takes it
    (see ConvCore.pool_fold): no pooled tensor, 5.32 -> 5.25 ms per batch of 32 slices of 512 x 512.  Training plans
    keep the pool and the 1x1 layer: there the 3x3 form's nine-tap data and weight gradients cost more than the two
    pool launches they replace (measured 7.89 -> 8.07 ms per step; RIED_POOL_FOLD_TRAIN=1 turns it on, the parity
    tests pass either way)."""
    N, H, W, _ = xin.shape
    if POOL_AFTER and not ctx.grads and ctx.st is not None and not ctx.per_image:
        # forward-only plans: the 1x1 convolution first (in the block's sibling group when there is one), the pool on
        # its cout channels.  32 slices of 512 x 512: the five pool branches took 0.97 ms of kernel time as folded 3x3
        # convolutions (one of them the longest branch of its block); see ConvBN(post_pool=True).
        seq.add(ConvBN(ctx, prefix + ".branch_pool", cin, cout, 1, 1, 0, xin, out_slice, accumulate=True,
                       post_pool=True))
        return
    if POOL_FOLD and cin % 64 == 0 and (not ctx.grads or os.environ.get("RIED_POOL_FOLD_TRAIN")):
        seq.add(ConvBN(ctx, prefix + ".branch_pool", cin, cout, 3, 1, 1, xin, out_slice, accumulate=True,
                       pool_fold=True))
        return
    pooled = Slot(ctx.act(N, H, W, cin))
    seq.add(Pool(ctx, ops.POOL_AVG3, xin, pooled))
    _chain(ctx, seq, prefix, [("branch_pool", cin, cout, 1, 1, 0)], pooled, out_slice, accumulate=True)


def _residual_head(ctx, seq, prefix, xin: Slot, out: Slot, cin, cout, has_ds):
    if has_ds:   # identity_downsample = Conv1x1 + BN + ReLU (src/rinet.py:99-103)
        seq.add(ConvBN(ctx, prefix + ".identity_downsample", cin, cout, 1, 1, 0, xin, out,
                       conv_key=prefix + ".identity_downsample.0", bn_key=prefix + ".identity_downsample.1"))
    else:
        assert cin == cout
        seq.add(Identity(xin, out))


def inception_a(ctx, prefix, xin: Slot, out: Slot, cin, has_ds) -> Seq:
    """src/rinet.py:207-252: out = cat(32|32|48|16) + identity."""
    s = Seq()
    grp = open_input_grad_group(ctx, xin)
    _residual_head(ctx, s, prefix, xin, out, cin, 128, has_ds)
    with _Lane(ctx, 0) as b0:      # the longest branch on the caller's stream
        _chain(ctx, b0, prefix, [("branch3x3dbl_1", cin, 32, 1, 1, 0), ("branch3x3dbl_2", 32, 48, 3, 1, 1),
                                 ("branch3x3dbl_3", 48, 48, 3, 1, 1)], xin, out.sub(64, 112), accumulate=True)
    with _Lane(ctx, 1) as b1:
        _chain(ctx, b1, prefix, [("branch3x3_1", cin, 16, 1, 1, 0), ("branch3x3_2", 16, 32, 3, 1, 1)], xin,
               out.sub(32, 64), accumulate=True)
    with _Lane(ctx, 2) as b2:
        _chain(ctx, b2, prefix, [("branch1x1", cin, 32, 1, 1, 0)], xin, out.sub(0, 32), accumulate=True)
    with _Lane(ctx, 3) as b3:
        _pool_branch(ctx, b3, prefix, xin, out.sub(112, 128), cin, 16)
    s.add(Par(ctx, [b0, b1, b2, b3]))
    close_input_grad_group(grp, s)
    return s


def inception_c(ctx, prefix, xin: Slot, out: Slot, cin, c7, has_ds) -> Seq:
    """src/rinet.py:293-347: out = cat(64|64|64|64) + identity."""
    s = Seq()
    grp = open_input_grad_group(ctx, xin)
    _residual_head(ctx, s, prefix, xin, out, cin, 256, has_ds)
    h, v = ((1, 7), 1, (0, 3)), ((7, 1), 1, (3, 0))
    with _Lane(ctx, 0) as b0:
        _chain(ctx, b0, prefix, [("branch7x7dbl_1", cin, c7, 1, 1, 0), ("branch7x7dbl_2", c7, c7, *v),
                                 ("branch7x7dbl_3", c7, c7, *h), ("branch7x7dbl_4", c7, c7, *v),
                                 ("branch7x7dbl_5", c7, 64, *h)], xin, out.sub(128, 192), accumulate=True)
    with _Lane(ctx, 1) as b1:
        _chain(ctx, b1, prefix, [("branch7x7_1", cin, c7, 1, 1, 0), ("branch7x7_2", c7, c7, *h),
                                 ("branch7x7_3", c7, 64, *v)], xin, out.sub(64, 128), accumulate=True)
    with _Lane(ctx, 2) as b2:
        _chain(ctx, b2, prefix, [("branch1x1", cin, 64, 1, 1, 0)], xin, out.sub(0, 64), accumulate=True)
    with _Lane(ctx, 3) as b3:
        _pool_branch(ctx, b3, prefix, xin, out.sub(192, 256), cin, 64)
    s.add(Par(ctx, [b0, b1, b2, b3]))
    close_input_grad_group(grp, s)
    return s


def inception_e(ctx, prefix, xin: Slot, out: Slot, cin, has_ds) -> Seq:
    """src/rinet.py:389-445: out = cat(160 | 192,192 | 192,192 | 96) + identity."""
    s = Seq()
    grp = open_input_grad_group(ctx, xin)
    _residual_head(ctx, s, prefix, xin, out, cin, 1024, has_ds)
    N, H, W, _ = xin.shape
    with _Lane(ctx, 0) as b0:
        bd = _chain(ctx, b0, prefix, [("branch3x3dbl_1", cin, 224, 1, 1, 0), ("branch3x3dbl_2", 224, 192, 3, 1, 1)],
                    xin, Slot(ctx.act(N, H, W, 192)))
        _chain(ctx, b0, prefix, [("branch3x3dbl_3a", 192, 192, (1, 3), 1, (0, 1))], bd, out.sub(544, 736),
               accumulate=True)
        _chain(ctx, b0, prefix, [("branch3x3dbl_3b", 192, 192, (3, 1), 1, (1, 0))], bd, out.sub(736, 928),
               accumulate=True)
    with _Lane(ctx, 1) as b1:
        b3 = _chain(ctx, b1, prefix, [("branch3x3_1", cin, 192, 1, 1, 0)], xin, Slot(ctx.act(N, H, W, 192)))
        _chain(ctx, b1, prefix, [("branch3x3_2a", 192, 192, (1, 3), 1, (0, 1))], b3, out.sub(160, 352),
               accumulate=True)
        _chain(ctx, b1, prefix, [("branch3x3_2b", 192, 192, (3, 1), 1, (1, 0))], b3, out.sub(352, 544),
               accumulate=True)
    with _Lane(ctx, 2) as b2:
        _chain(ctx, b2, prefix, [("branch1x1", cin, 160, 1, 1, 0)], xin, out.sub(0, 160), accumulate=True)
    with _Lane(ctx, 3) as b4:
        _pool_branch(ctx, b4, prefix, xin, out.sub(928, 1024), cin, 96)
    s.add(Par(ctx, [b0, b1, b2, b4]))
    close_input_grad_group(grp, s)
    return s


def _grid_tail(ctx, s, prefix, xin: Slot, out_slice: Slot, cin, cs, decoder):
    N, H, W, _ = xin.shape
    if decoder:   # sample = conv1x1(no bias) + bilinear x2, align_corners (src/rinet.py:264-267, 359-362)
        mid = Slot(ctx.act(N, H, W, cs))
        s.add(ConvPlain(ctx, prefix + ".sample.0.weight", cin, cs, xin, mid))
        s.add(Upsample(ctx, mid, out_slice))
    else:
        s.add(Pool(ctx, ops.POOL_MAX3S2, xin, out_slice))


def inception_b(ctx, prefix, xin: Slot, out: Slot, cin, decoder) -> Seq:
    """src/rinet.py:254-291: cat(64 | 64 | maxpool(cin) or sample(128))."""
    k = 4 if decoder else 3
    grp = open_input_grad_group(ctx, xin)      # (decoder blocks: the siblings are 1x1 transposed convolutions)
    with _Lane(ctx, 0) as b0:
        _chain(ctx, b0, prefix, [("branch3x3dbl_1", cin, 32, 1, 1, 0), ("branch3x3dbl_2", 32, 48, 3, 1, 1),
                                 ("branch3x3dbl_3", 48, 64, k, 2, 1)], xin, out.sub(64, 128), deconv=decoder)
    with _Lane(ctx, 1) as b1:
        _chain(ctx, b1, prefix, [("branch3x3_1", cin, 32, 1, 1, 0), ("branch3x3_2", 32, 64, k, 2, 1)], xin,
               out.sub(0, 64), deconv=decoder)
    cs = 128 if decoder else cin
    with _Lane(ctx, 2) as b2:
        _grid_tail(ctx, b2, prefix, xin, out.sub(128, 128 + cs), cin, cs, decoder)
    s = Seq([Par(ctx, [b0, b1, b2])])
    close_input_grad_group(grp, s)
    return s


def inception_d(ctx, prefix, xin: Slot, out: Slot, cin, decoder) -> Seq:
    """src/rinet.py:349-387: cat(128 | 128 | maxpool(cin) or sample(256))."""
    k = 4 if decoder else 3
    grp = open_input_grad_group(ctx, xin)      # (decoder blocks: the siblings are 1x1 transposed convolutions)
    with _Lane(ctx, 0) as b0:
        _chain(ctx, b0, prefix, [("branch7x7x3_1", cin, 128, 1, 1, 0), ("branch7x7x3_2", 128, 128, (1, 7), 1, (0, 3)),
                                 ("branch7x7x3_3", 128, 128, (7, 1), 1, (3, 0)),
                                 ("branch7x7x3_4", 128, 128, k, 2, 1)], xin, out.sub(128, 256), deconv=decoder)
    with _Lane(ctx, 1) as b1:
        _chain(ctx, b1, prefix, [("branch3x3_1", cin, 64, 1, 1, 0), ("branch3x3_2", 64, 128, k, 2, 1)], xin,
               out.sub(0, 128), deconv=decoder)
    cs = 256 if decoder else cin
    with _Lane(ctx, 2) as b2:
        _grid_tail(ctx, b2, prefix, xin, out.sub(256, 256 + cs), cin, cs, decoder)
    s = Seq([Par(ctx, [b0, b1, b2])])
    close_input_grad_group(grp, s)
    return s


# --------------------------------------------------------------------------- generator
class Generator:
    """RIGAN (src/rinet.py:34-92) for a fixed (N, H, W).  x and the output are fp32 (N,1,H,W)."""

    # first top-level module of each gradient bucket, in forward (= flat parameter) order; the backward pass
    # completes the buckets last to first
    BUCKET_STARTS = ("conv1", "layer_block_3")
    # top-level modules in forward order (src/rinet.py:55-70); out_conv's gradient is the first one backward computes
    TOP_LEVEL = ("conv1", "conv2", "layer_block_1", "reduce_grid_1", "layer_block_2", "reduce_grid_2", "layer_block_3",
                 "expand_grid_1", "layer_block_4", "expand_grid_2", "layer_block_5", "deconv1", "deconv2", "out_conv")
    # (Measured without data parallelism: launching each bucket's deferred weight gradients on a side stream, beside
    # the remaining layers' small kernels, makes the step 0.09 ms slower; they stay at the end of the pass.)

    def __init__(self, state: State, N: int, H: int, W: int, grads=True, training=True, device="cuda",
                 out: Optional[torch.Tensor] = None, per_image_stats: bool = False, grad_buckets: bool = False,
                 bn_sync=None):
        """out: where the (N,1,H,W) fp32 result goes (default: an own buffer).
        grad_buckets: finish the parameter gradients in two buckets - bottleneck + decoder (81 % of the parameters,
        complete after the first half of the backward pass), then the encoder - instead of all at the end of
        backward(), so that a data-parallel step can exchange the first while the second is being computed: iterate
        backward_buckets() and read bucket_ranges.
        per_image_stats (forward-only, train-mode BatchNorm): every image of the batch is normalised with its own
        statistics, i.e. the batch is N independent batch-1 forwards - what the reference's evaluation loop
        computes (src/test.py:107-131: batch-1 loader, the model left in train mode).  The running statistics are
        not updated.  Needs H, W divisible by 256 (the smallest maps must consist of whole pixel tiles)."""
        assert H % 16 == 0 and W % 16 == 0, "RIGAN needs H, W divisible by 16 (skip connections must line up)"
        self.st = state
        self.ctx = ctx = Ctx(state, device, grads, training)
        if bn_sync is not None:      # (see Ctx.bn_sync)
            if not (grads and training and BN_FOLD):
                raise ValueError("synchronised BatchNorm: training plans with folded statistics only")
            ctx.bn_sync = bn_sync
        # every weight gradient of this network runs in the grouped launches: the grid is shared by all layers, so a
        # layer needs far fewer blocks of its own (fewer split-K slabs to write and reduce) than a launch per layer
        ctx.splitk_target = int(os.environ.get("RIED_SPLITK_TARGET_G", "0"))
        ctx.splitk_minpix = int(os.environ.get("RIED_SPLITK_MINPIX_G", "0"))
        if per_image_stats:
            if grads or not training or H % 256 or W % 256:
                raise ValueError("per-image statistics: forward-only train-mode plans on H, W divisible by 256")
            ctx.per_image = True
        self.N, self.H, self.W = N, H, W
        h2, w2, h4, w4, h8, w8, h16, w16 = H // 2, W // 2, H // 4, W // 4, H // 8, W // 8, H // 16, W // 16
        self.cat2 = Slot(ctx.act(N, h2, w2, 96))        # cat(conv1 (32), deconv1 (64))   src/rinet.py:89
        self.cat1 = Slot(ctx.act(N, h4, w4, 192))       # cat(conv2 (64), layer_block_5 (128))   src/rinet.py:88
        self.conv1 = ConvBNC1(ctx, "conv1", 32, N, H, W, self.cat2.sub(0, 32))
        body = Seq()
        self.bucket_first = [0]        # index into body.layers of the first layer of buckets 1, 2, ... (bucket 0: conv1)
        bucketed = grads and grad_buckets
        # [lo, hi) of the flat parameter / gradient buffer per bucket.  The flat layout follows the key order of the
        # state dict the caller passed: with the reference's order (module definition order) each bucket's parameters
        # are one contiguous range; with any other order (sorted keys, say) they may interleave, and then the plan
        # keeps a single bucket rather than exchange a range that holds unfinished gradients.
        self.bucket_ranges = [(0, state.flat.numel())]
        if bucketed:
            first_late = self.TOP_LEVEL.index(self.BUCKET_STARTS[1])
            late = set(self.TOP_LEVEL[first_late:])
            spans = [[None, 0], [None, 0]]
            for k, off in state.offs.items():
                b = 1 if k.split(".")[0] in late else 0
                end = off + (state.p[k].numel() + 3) // 4 * 4
                spans[b][0] = off if spans[b][0] is None else min(spans[b][0], off)
                spans[b][1] = max(spans[b][1], end)
            (lo0, hi0), (lo1, hi1) = spans
            if lo0 == 0 and lo1 is not None and hi0 <= lo1 and hi1 == state.flat.numel():
                self.bucket_ranges = [(0, lo1), (lo1, hi1)]
            else:
                bucketed = False
        starts_at = self.BUCKET_STARTS if bucketed else ()

        def bucket(name):
            if name in starts_at[1:]:
                ctx.set_bucket(starts_at.index(name))
                self.bucket_first.append(len(body.layers))
        c2 = self.cat1.sub(0, 64)
        body.add(ConvBN(ctx, "conv2", 32, 64, 3, 2, 1, self.cat2.sub(0, 32), c2))
        a1 = Slot(ctx.act(N, h4, w4, 128)); body.add(inception_a(ctx, "layer_block_1.0", c2, a1, 64, True))
        a2 = Slot(ctx.act(N, h8, w8, 256)); body.add(inception_b(ctx, "reduce_grid_1", a1, a2, 128, False))
        a3 = Slot(ctx.act(N, h8, w8, 256)); body.add(inception_c(ctx, "layer_block_2.0", a2, a3, 256, 64, False))
        a4 = Slot(ctx.act(N, h16, w16, 512)); body.add(inception_d(ctx, "reduce_grid_2", a3, a4, 256, False))
        bucket("layer_block_3")
        a5 = Slot(ctx.act(N, h16, w16, 1024)); body.add(inception_e(ctx, "layer_block_3.0", a4, a5, 512, True))
        a6 = Slot(ctx.act(N, h8, w8, 512)); body.add(inception_d(ctx, "expand_grid_1", a5, a6, 1024, True))
        a7 = Slot(ctx.act(N, h8, w8, 256)); body.add(inception_c(ctx, "layer_block_4.0", a6, a7, 512, 64, True))
        a8 = Slot(ctx.act(N, h4, w4, 256)); body.add(inception_b(ctx, "expand_grid_2", a7, a8, 256, True))
        body.add(inception_a(ctx, "layer_block_5.0", a8, self.cat1.sub(64, 192), 256, True))
        # Forward-only plans with batch statistics: deconv1 stops at its statistics and leaves its RAW output in its
        # slice of cat2; deconv2 - its one consumer, on the halo kernel whose input tile serves all sixteen taps -
        # normalises those 64 channels as it loads them (ConvCore.xf_mixed): the pass that read and rewrote the
        # 64-channel half-resolution tensor (126 us of a 4.9 ms batch of 32 slices of 512 x 512) is gone.
        d1_raw = (XF_FUSE and D4_XF and not grads and training and not ctx.per_image and ctx.st is not None
                  and h2 % 16 == 0 and w2 % 16 == 0)
        d1 = body.add(ConvBN(ctx, "deconv1", 192, 64, 4, 2, 1, self.cat1, self.cat2.sub(32, 96), deconv=True,
                             defer_apply=d1_raw, raw_view=self.cat2.sub(32, 96).v if d1_raw else None))
        self.hlast = Slot(ctx.act(N, H, W, 32))
        # forward-only plans: out_conv reads deconv2's raw output and normalises on the fly (hlast stays unwritten)
        self.last = body.add(ConvBN(ctx, "deconv2", 96, 32, 4, 2, 1, self.cat2, self.hlast, deconv=True,
                                    defer_apply=not grads))
        if d1_raw:
            ident = torch.cat([torch.zeros(96), torch.ones(96)]).to(device=device, dtype=F32)     # mean 0, invstd 1
            self.last.core.xf_mixed = (d1, 32, ident, torch.ones(96, dtype=F32, device=device),
                                       torch.zeros(96, dtype=F32, device=device))
        self.body = body
        self.tops = {"conv1": self.cat2.sub(0, 32), "conv2": c2, "layer_block_1": a1, "reduce_grid_1": a2,
                     "layer_block_2": a3, "reduce_grid_2": a4, "layer_block_3": a5, "expand_grid_1": a6,
                     "layer_block_4": a7, "expand_grid_2": a8, "layer_block_5": self.cat1.sub(64, 192),
                     "deconv1": self.cat2.sub(32, 96), "deconv2": self.hlast}
        self.out = torch.empty(N, 1, H, W, dtype=F32, device=device) if out is None else out
        assert self.out.shape == (N, 1, H, W) and self.out.dtype == F32 and self.out.is_contiguous()
        ctx.finalize()
        self.x = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape == (self.N, 1, self.H, self.W) and x.dtype == F32 and x.is_contiguous()
        self.x = x
        self.ctx.begin_forward()
        self.conv1.forward(x)
        self.body.forward()
        if self.last.defer_apply:
            pi = self.last.per_image
            ops.out_conv_fwd(x, self.last.raw, self.st.p["out_conv.weight"], self.out,
                             bn=(self.last.img_save if pi else self.last.save, self.st.p["deconv2.bn.weight"],
                                 self.st.p["deconv2.bn.bias"]), bn_images=self.N if pi else 0)
        else:
            ops.out_conv_fwd(x, self.hlast.v, self.st.p["out_conv.weight"], self.out)
        return self.out

    def backward(self, dout: torch.Tensor):
        """dout: fp32 (N,1,H,W) = d loss / d output.  Fills state.g (which must be zero on entry)."""
        for _ in self.backward_buckets(dout):
            pass

    def backward_buckets(self, dout: torch.Tensor):
        """backward() as a generator: yields b each time bucket b of the parameter gradients
        (state.flat_grad[bucket_ranges[b]]) is complete on the current stream - the last bucket first."""
        assert dout.shape == self.out.shape and dout.dtype == F32 and dout.is_contiguous()
        self.ctx.zero_backward()
        ops.out_conv_bwd(self.x, self.hlast.v, self.st.p["out_conv.weight"], dout, self.hlast.g,
                         self.st.g["out_conv.weight"])
        layers = self.body.layers
        for i in reversed(range(len(layers))):
            layers[i].backward()
            if i in self.bucket_first[1:]:
                b = self.bucket_first.index(i)
                self.ctx.end_backward(b)
                yield b
        self.conv1.backward()
        self.ctx.end_backward(0)
        yield 0


# --------------------------------------------------------------------------- discriminator
class SNConv:
    """SNConv2d (src/rinet.py:131-148): spectral-norm 3x3 conv pad 1 with bias, LeakyReLU(0.2)."""

    def __init__(self, ctx: Ctx, prefix: str, cin, cout, stride, xin: Optional[Slot], out: Slot, first=False):
        self.ctx, self.prefix, self.cin, self.cout, self.stride, self.first = ctx, prefix, cin, cout, stride, first
        self.out = out
        dev = ctx.device
        self.K = cin * 9
        self.sigma = torch.ones(1, dtype=F32, device=dev)
        N, Ho, Wo, _ = out.shape
        self.dwsn = torch.empty(cout * self.K, dtype=F32, device=dev) if ctx.grads else None
        if not first:
            self.core = ConvCore(ctx, prefix + ".weight_orig", cin, cout, 3, stride, 1, False, xin, True,
                                 sigma=self.sigma, wgrad_out=self.dwsn)
        if ctx.grads:
            ctx.need_tmp(N * Ho * Wo * cout)
            ctx.need_f32(ops.colsum_scratch_floats(cout))
            if first:
                ctx.need_f32(ops.conv_c1_wgrad_scratch_floats(cout))
                if C1_FOLD and cout == 64 and stride == 1 and Ho % 32 == 0 and Wo % 32 == 0:
                    ctx.need_f32(ops.conv2d_s2_dgrad_c1fold_scratch_floats(N, Ho // 2, Wo // 2))
                    self.foldable = True
            out.act.fuse_slope = 0.2          # whoever writes out.g may apply this layer's LeakyReLU backward
            if first and SIGN_MASK and cout % 8 == 0 and (out.c0, out.c1, out.n0, out.n1) == (0, cout, 0, N):
                out.act.signs = torch.empty(N * Ho * Wo, cout // 8, dtype=torch.uint8, device=dev)
            if not first and xin.act.fuse_slope is not None:
                _, H, W, _ = xin.shape
                xin.act.fused = _Fused(ctx, N, H, W, cout, cin, 3, stride, stride)
        st = ctx.st
        ctx.sn.add(st.p[prefix + ".weight_orig"], st.b[prefix + ".weight_u"], st.b[prefix + ".weight_v"], self.sigma,
                   cout, self.K, self.dwsn, st.g[prefix + ".weight_orig"] if ctx.grads else None)
        self.x_img: Optional[torch.Tensor] = None
        # Set on the layer above a foldable first layer (by Discriminator): when a backward pass wants parameter
        # gradients only, that layer's data gradient forms this layer's weight gradient in its epilogue and never
        # stores the gradient of this layer's output (ops.conv2d_s2_dgrad_c1fold).
        self.foldable = getattr(self, "foldable", False)
        self.below: Optional["SNConv"] = None
        self._folded = False

    def forward(self, x_img: Optional[torch.Tensor] = None, n0: int = 0, n1: Optional[int] = None):
        """Images [n0, n1) only (the layer has no statistics across the batch: halves may run at different times)."""
        st, p = self.ctx.st, self.prefix
        n1 = self.out.shape[0] if n1 is None else n1
        out = self.out.images(n0, n1).v
        if self.first:
            self.x_img = x_img
            sg = self.out.act.signs
            if sg is not None:
                hw = out.H * out.W
                sg = sg[n0 * hw:n1 * hw]
            ops.conv_c1_fwd(x_img[n0:n1], st.p[p + ".weight_orig"], out, self.stride, sigma=self.sigma,
                            bias=st.p[p + ".bias"], act=2, signs=sg)
        else:
            self.core.conv(out, x=self.core.xin.images(n0, n1).v, bias=st.p[p + ".bias"], act=2)

    def backward(self, n0: int, n1: int, wgrad: bool, dimg: Optional[torch.Tensor] = None):
        """Images [n0, n1) only.  wgrad: parameter gradients; dimg (first layer): fp32 image gradient, +=."""
        ctx, st, p = self.ctx, self.ctx.st, self.prefix
        out = self.out.images(n0, n1)
        N, Ho, Wo, C = out.shape
        fz = self.out.act.fused
        if fz is not None:       # out.g already holds the pre-activation gradient; its column sums are in fz.stats
            dpre = out.g
            if wgrad:
                assert N == self.out.shape[0], "bias column sums cover the whole batch"
                ops.stat_rows_sum(fz.stats, fz.rows, C, st.g[p + ".bias"])
        else:
            dpre = ctx.tmp_view(N, Ho, Wo, C)
            ops.ew(ops.EW_ACTBWD, out.g, dpre, y=out.v, slope=0.2)
            if wgrad:
                ops.colsum(dpre, st.g[p + ".bias"], ctx.f32)
        if wgrad:
            if self.first and self._folded:      # the layer above has already added this pass's sums to dwsn
                self._folded = False
            elif self.first:
                self.dwsn.zero_()
                ops.conv_c1_wgrad(self.x_img[n0:n1], dpre, self.stride, self.dwsn, ctx.f32)
            else:
                self.core.wgrad(dpre, x=self.core.xin.images(n0, n1).v)
        if self.first:
            if dimg is not None:
                ops.conv_c1_dgrad(dpre, st.p[p + ".weight_orig"], dimg, sigma=self.sigma, accumulate=True)
        else:
            xin = self.core.xin.images(n0, n1)
            xf = self.core.xin.act.fused
            sg = self.core.xin.act.signs
            lo = self.below
            if (xf is not None and lo is not None and wgrad and dimg is None and self.stride == 2 and
                    (xin.c0, xin.c1) == (0, 64) and xin.shape[0] == lo.out.shape[0]):
                hw = xin.shape[1] * xin.shape[2]
                lo.dwsn.zero_()
                ops.conv2d_s2_dgrad_c1fold(dpre, self.core.wpt, xin.v if sg is None else sg[xin.n0 * hw:xin.n1 * hw],
                                           self.core.xin.act.fuse_slope, xf.stats, lo.x_img[n0:n1], lo.dwsn, ctx.f32)
                lo._folded = True
            elif xf is not None and sg is not None:
                c, hw = self.core, xin.shape[1] * xin.shape[2]
                assert (xin.c0, xin.c1) == (0, sg.shape[1] * 8)
                ops.conv2d_signs(dpre, c.wpt, xin.g, c.kh, c.kw, c.sh, c.sw, c.ph, c.pw, sg[xin.n0 * hw:xin.n1 * hw],
                                 transposed=True, mask_slope=c.xin.act.fuse_slope, stats=xf.stats if wgrad else None,
                                 stats_post=True)
            elif xf is not None:
                self.core.dgrad(dpre, xin.g, accumulate=False, mask=xin.v, mask_slope=self.core.xin.act.fuse_slope,
                                stats=xf.stats if wgrad else None, stats_post=True)
            else:
                self.core.dgrad(dpre, xin.g, accumulate=False)


class Attn:
    """Self_Attn (src/rinet.py:150-181) on NHWC tokens: the q|k|v projections are one 1x1 conv with
    concatenated output channels; energy, softmax and the value product are batched MFMA GEMMs."""

    def __init__(self, ctx: Ctx, prefix: str, C: int, xin: Slot, out: Slot):
        self.ctx, self.prefix, self.C, self.d = ctx, prefix, C, C // 8
        self.xin, self.out = xin, out
        N, H, W, _ = xin.shape
        self.N, self.n = N, H * W
        dev = ctx.device
        self.cq = 2 * self.d + C
        self.qkv = Slot(ctx.act(N, H, W, self.cq))
        self.wp = torch.empty(self.cq * C, dtype=torch.bfloat16, device=dev)
        st = ctx.st                  # State lays q, k, v weights (and biases) out back to back: views, no copies
        ow, ob = st.offs[prefix + ".query_conv.weight"], st.offs[prefix + ".query_conv.bias"]
        assert st.offs[prefix + ".value_conv.weight"] == ow + 2 * self.d * C
        assert st.offs[prefix + ".value_conv.bias"] == ob + 2 * self.d
        self.wcat = st.flat[ow:ow + self.cq * C].view(self.cq, C)           # [q; k; v] projection weights
        self.bias = st.flat[ob:ob + self.cq]
        ctx.add_pack(self.wcat, self.wp, self.cq, 1, 1, C, C, 1, 1, 1)
        self.fused = FUSED_ATTN and ops.attn_fused_supported(self.n, self.d, C)    # one kernel each way, no n x n tensors
        self.O = ops.new_view(N, H, W, C, dev)
        if self.fused:
            self.lse = torch.empty(N * self.n, dtype=F32, device=dev)
        else:
            self.S = torch.empty(N * self.n * self.n, dtype=F32, device=dev)
            self.P = torch.empty(N * self.n * self.n, dtype=torch.bfloat16, device=dev)
        if ctx.grads:
            self.wpt = torch.empty(self.cq * C, dtype=torch.bfloat16, device=dev)
            if not self.fused:
                self.dS = torch.empty(N * self.n * self.n, dtype=torch.bfloat16, device=dev)
                self.dO = ops.new_view(N, H, W, C, dev)
            self.dw = st.flat_grad[ow:ow + self.cq * C]
            self.db = st.flat_grad[ob:ob + self.cq]
            ctx.add_pack(self.wcat, self.wpt, C, 1, 1, self.cq, 1, 1, 1, C)   # [C][1][1][cq] for the data gradient
            kp = N * self.n
            self.splitk = _splitk(ops.conv2d_wgrad_tiles(C, self.cq), kp, whole_rounds=True)
            self.slab_off = ctx.alloc_slab(self.splitk * self.cq * C)
            ctx.add_wgrad_reduce(self.slab_off, self.dw, self.splitk, 1, self.cq, C, 1, C, 1)
            ctx.need_f32(ops.colsum_scratch_floats(self.cq))
            if xin.act.fuse_slope is not None:
                xin.act.fused = _Fused(ctx, N, H, W, self.cq, C, 1, 1, 1)

    def _names(self):
        p = self.prefix
        return [(p + ".query_conv", 0, self.d), (p + ".key_conv", self.d, 2 * self.d),
                (p + ".value_conv", 2 * self.d, self.cq)]

    def forward(self, n0: int = 0, n1: Optional[int] = None):
        st, C, d, n, N = self.ctx.st, self.C, self.d, self.n, self.N
        if self.fused:        # per image: any range of the batch
            n1 = N if n1 is None else n1
            x, qkv = self.xin.images(n0, n1).v, self.qkv.images(n0, n1).v
            ops.conv2d(x, self.wp, qkv, 1, 1, 1, 1, 0, 0, bias=self.bias)
            ops.attn_fwd(qkv, x, self.out.images(n0, n1).v, self.O.images(n0, n1), self.lse[n0 * n:], st.p[self.prefix + ".gamma"], d)
            return
        assert n0 == 0 and n1 in (None, N), "the unfused attention path runs the whole batch"
        x, qkv = self.xin.v, self.qkv.v
        ops.conv2d(x, self.wp, qkv, 1, 1, 1, 1, 0, 0, bias=self.bias)
        q, k, v = qkv.ptr, qkv.ptr + 2 * d, qkv.ptr + 4 * d
        ld, bs = qkv.ld, n * qkv.ld
        # S[b][i][j] = q_i . k_j : m = j (A = k), n = i (B = q)
        if ops.attn_scores_softmax_supported(n, d):
            ops.attn_scores_softmax(qkv, d, self.P)
        else:
            ops.bgemm(k, ld, bs, 0, q, ld, bs, 0, n, n, d, N, self.S.data_ptr(), n, n * n, out_f32=True)
            ops.softmax_fwd(self.S, self.P, N * n, n)
        # O[b][i][c] = sum_j P[i][j] v[j][c] : m = c (A = v stored [j][c]), n = i (B = P)
        ops.bgemm(v, ld, bs, 1, self.P.data_ptr(), n, n * n, 0, C, n, n, N, self.O.ptr, C, n * C)
        ops.ew(ops.EW_AXPY, self.O, self.out.v, y=x, s=st.p[self.prefix + ".gamma"])

    def backward(self, n0: int, n1: int, wgrad: bool):
        st, C, d, n = self.ctx.st, self.C, self.d, self.n
        N = n1 - n0
        out, xin, qkv = self.out.images(n0, n1), self.xin.images(n0, n1), self.qkv.images(n0, n1)
        O = self.O.images(n0, n1)
        gamma = st.p[self.prefix + ".gamma"]
        if out.g.ptr != xin.g.ptr:      # residual path (first writer of xin.g); free when the two views alias
            ops.ew(ops.EW_COPY, out.g, xin.g)
        if self.fused:
            ops.attn_bwd(qkv.v, qkv.g, out.g, O, self.lse.data_ptr() + 4 * n0 * n, gamma,
                         st.g[self.prefix + ".gamma"] if wgrad else None, d)
            self._backward_projection(xin, qkv, wgrad)
            return
        dO = self.dO.images(n0, n1)
        if wgrad:
            ops.dot(out.g, O, 1.0, st.g[self.prefix + ".gamma"])
        ops.ew(ops.EW_SCALE, out.g, dO, s=gamma)
        ld, bs = qkv.v.ld, n * qkv.v.ld
        q, k, v = qkv.v.ptr, qkv.v.ptr + 2 * d, qkv.v.ptr + 4 * d
        dq, dk, dv = qkv.g.ptr, qkv.g.ptr + 2 * d, qkv.g.ptr + 4 * d
        off = n0 * n * n
        P = self.P.data_ptr() + 2 * off
        dP = self.S.data_ptr() + 4 * off                                    # S is dead after the softmax: reuse
        dS = self.dS.data_ptr() + 2 * off
        # dP[i][j] = sum_c dO[i][c] v[j][c] : m = j (A = v [j][c]), n = i (B = dO)
        ops.bgemm(v, ld, bs, 0, dO.ptr, C, n * C, 0, n, n, C, N, dP, n, n * n, out_f32=True)
        # dv[j][c] = sum_i P[i][j] dO[i][c] : m = c (A = dO stored [i][c]), n = j (B = P stored [i][j])
        ops.bgemm(dO.ptr, C, n * C, 1, P, n, n * n, 1, C, n, n, N, dv, ld, bs)
        ops.softmax_bwd(self.S[off:], self.P[off:], self.dS[off:], N * n, n)
        # dq[i][e] = sum_j dS[i][j] k[j][e] : m = e (A = k stored [j][e]), n = i (B = dS)
        ops.bgemm(k, ld, bs, 1, dS, n, n * n, 0, d, n, n, N, dq, ld, bs)
        # dk[j][e] = sum_i dS[i][j] q[i][e] : m = e (A = q stored [i][e]), n = j (B = dS stored [i][j])
        ops.bgemm(q, ld, bs, 1, dS, n, n * n, 1, d, n, n, N, dk, ld, bs)
        self._backward_projection(xin, qkv, wgrad)

    def _backward_projection(self, xin, qkv, wgrad: bool):
        """Backward of the fused [q; k; v] 1x1 convolution from qkv.g."""
        if wgrad:             # bias and weight gradients land directly in the state's (zeroed) gradient buffer
            ops.colsum(qkv.g, self.db, self.ctx.f32)
            ops.conv2d_wgrad(xin.v, qkv.g, 1, 1, 1, 1, 0, 0, self.ctx.slab_at(self.slab_off), self.splitk)
        xf = self.xin.act.fused
        if xf is not None:
            ops.conv2d(qkv.g, self.wpt, xin.g, 1, 1, 1, 1, 0, 0, transposed=True, accumulate=True, mask=xin.v,
                       mask_slope=self.xin.act.fuse_slope, stats=xf.stats if wgrad else None, stats_post=True)
        else:
            ops.conv2d(qkv.g, self.wpt, xin.g, 1, 1, 1, 1, 0, 0, transposed=True, accumulate=True)


class Discriminator:
    """ImageDiscriminator (src/rinet.py:10-32) for a fixed batch of N one-channel fp32 images."""

    def __init__(self, state: State, N: int, H: int, W: int, grads=True, training=True, device="cuda"):
        assert H % 16 == 0 and W % 16 == 0
        self.st = state
        self.ctx = ctx = Ctx(state, device, grads, training)
        ctx.splitk_whole_rounds = True       # every weight gradient here is a launch of its own
        self.N, self.H, self.W = N, H, W
        p = "discriminator_net."
        a0 = Slot(ctx.act(N, H, W, 64))
        a1 = Slot(ctx.act(N, H // 2, W // 2, 128))
        a2 = Slot(ctx.act(N, H // 4, W // 4, 256))
        a3 = Slot(ctx.act(N, H // 4, W // 4, 256))
        a4 = Slot(ctx.act(N, H // 8, W // 8, 512))
        a5 = Slot(ctx.act(N, H // 8, W // 8, 512))
        a6 = Slot(ctx.act(N, H // 16, W // 16, 512))
        if grads:
            # An attention layer's output gradient is dead once its residual path has been added to the input
            # gradient, so the two share memory: the residual copy disappears and the layer's last data gradient
            # accumulates in place.
            a3.act.g = a2.act.g
            a5.act.g = a4.act.g
        self.acts = [a0, a1, a2, a3, a4, a5, a6]
        self.layers = [
            SNConv(ctx, p + "0.conv2d", 1, 64, 1, None, a0, first=True),
            SNConv(ctx, p + "1.conv2d", 64, 128, 2, a0, a1),
            SNConv(ctx, p + "2.conv2d", 128, 256, 2, a1, a2),
            Attn(ctx, p + "3", 256, a2, a3),
            SNConv(ctx, p + "4.conv2d", 256, 512, 2, a3, a4),
            Attn(ctx, p + "5", 512, a4, a5),
            SNConv(ctx, p + "6.conv2d", 512, 512, 2, a5, a6),
        ]
        if self.layers[0].foldable:
            self.layers[1].below = self.layers[0]
        ctx.chain = True         # every gradient buffer has exactly one first writer that overwrites it
        ctx.finalize()
        self.out_features = 512 * (H // 16) * (W // 16)

    @property
    def out(self) -> Slot:
        return self.acts[-1]

    def forward(self, x_img: torch.Tensor) -> Slot:
        """x_img: fp32 (N,1,H,W).  Returns the last activation (N, H/16, W/16, 512) NHWC bf16."""
        self.prepare()
        return self.forward_images(x_img, 0, self.N)

    @property
    def splittable(self) -> bool:
        """True if forward_images() accepts a sub-range of the batch."""
        return all(l.fused for l in self.layers if isinstance(l, Attn))

    def prepare(self, power_iter: Optional[bool] = None):
        """The part of a forward pass that depends on the weights only: one power iteration per spectral-norm layer
        (src/rinet.py:138 runs it in every training-mode forward), sigma, and the packed W / sigma.
        power_iter=False: sigma and the packed weights from the vectors as they are (they already hold this
        forward's iteration: a resumed look-ahead plan)."""
        self.ctx.sn_sigmas(power_iter)
        self.ctx.begin_forward()

    def forward_images(self, x_img: torch.Tensor, n0: int, n1: int) -> Slot:
        """The layers on images [n0, n1) of x_img (N,1,H,W), after prepare().  No layer couples the images of a batch
        (no BatchNorm; attention is per image), so ranges may run at different times and on different streams."""
        assert x_img.shape == (self.N, 1, self.H, self.W) and x_img.dtype == F32 and x_img.is_contiguous()
        for i, l in enumerate(self.layers):
            if i == 0:
                l.forward(x_img, n0, n1)
            elif isinstance(l, SNConv):
                l.forward(None, n0, n1)
            else:
                l.forward(n0, n1)
        return self.acts[-1]

    def backward(self, n0: int, n1: int, wgrad: bool, dimg: Optional[torch.Tensor] = None):
        """The gradient wrt the last activation must already be in out.g for images [n0, n1); every other
        gradient buffer must be zero (call ctx.zero_backward() before writing out.g)."""
        for l in reversed(self.layers):
            if isinstance(l, SNConv):
                l.backward(n0, n1, wgrad, dimg)
            else:
                l.backward(n0, n1, wgrad)
        if wgrad:
            self.ctx.end_backward()
            self.ctx.sn_finish_wgrads()

    def output_nchw(self) -> torch.Tensor:
        """(N, 8192-like) fp32 in the reference's flattening order (C, H, W) (src/rinet.py:31)."""
        v = self.acts[-1].v
        y = torch.empty(v.N, v.C, v.H, v.W, dtype=F32, device=self.ctx.device)
        ops.nhwc_to_nchw(v, y)
        return y.view(v.N, -1)


# --------------------------------------------------------------------------- VGG16 taps
VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
VGG_TAPS = (3, 8, 15, 22, 29)


class Vgg:
    """Frozen VGG16 features[0:30] with the five post-ReLU taps of src/loss.py:71-87, run on the batch
    [prediction; target] (2B one-channel fp32 images, normalised on load as normalize_ does).  The three
    identical input channels of torch.cat([x,x,x]) (src/loss.py:181-182) are folded into the first layer's
    weights.  backward() propagates only through the prediction half and only to the data."""

    def __init__(self, weights: Dict[str, torch.Tensor], B: int, H: int, W: int, grads=True, device="cuda",
                 normalize=True):
        """normalize: apply normalize_ (src/loss.py:20-22, (x+1024)/4096) on load, as CompoundLoss does
        (src/loss.py:179-180); MPL feeds the raw images (src/loss.py:36,101-102)."""
        self.ctx = ctx = Ctx(None, device, grads, training=False)
        self.B, self.H, self.W = B, H, W
        # True (set by the training step): forward_pred_fused() compares with the target half's stored bf16 activations
        # and the target pass (forward(..., taps32=False)) writes no fp32 taps
        self.bf16_targets = False
        self.in_scale, self.in_shift = (1.0 / 4096.0, 0.25) if normalize else (1.0, 0.0)
        N = 2 * B
        dev = device
        self.w0 = weights["0.weight"].to(dev, F32).sum(1).reshape(64, 9).contiguous()     # fold 3 -> 1 channel
        self.b0 = weights["0.bias"].to(dev, F32).contiguous()
        self.layers = []      # ("conv", idx, ConvCore, bias, out Slot) | ("pool", Pool)
        self.taps: List[Slot] = []
        cur = Slot(ctx.act(N, H, W, 64))
        self.a0 = cur
        idx, cin, h, w = 2, 64, H, W
        cur_is_conv, cur_is_tap = True, False          # what produced `cur`; whether the loss seeds its gradient
        for v in VGG_CFG[1:]:
            if v == "M":
                nxt = Slot(ctx.act(N, h // 2, w // 2, cin))
                self.layers.append(("pool", Pool(ctx, ops.POOL_MAX2S2, cur, nxt), cur_is_tap))
                h, w = h // 2, w // 2
                idx += 1
                cur_is_conv, cur_is_tap = False, False
            else:
                nxt = Slot(ctx.act(N, h, w, v))
                core = ConvCore(ctx, f"{idx}.weight", cin, v, 3, 1, 1, False, cur, True, need_wgrad=False)
                cin_prev = cin
                wt = weights[f"{idx}.weight"].to(dev, F32).contiguous()
                ops.pack_weight(wt, core.wp, v, 3, 3, cin, cin * 9, 3, 1, 9)
                if grads:
                    ops.pack_weight(wt, core.wpt, cin, 3, 3, v, 9, 3, 1, cin * 9)
                cin = v
                idx += 2
                tap32 = None
                if idx - 1 in VGG_TAPS:        # taps are also kept in fp32: the loss is a small difference
                    tap32 = torch.empty(N, h, w, v, dtype=F32, device=dev)
                    self.taps.append((nxt, tap32))
                    core.mse_rows = ops.conv2d_stat_rows(B, h, w, cin_prev, v, 3, 3)      # the fused prediction pass
                    ctx.stats(core, "mse_stats", core.mse_rows * 2 * v)
                self.layers.append(("conv", core, weights[f"{idx - 2}.bias"].to(dev, F32).contiguous(), nxt, tap32,
                                    cur_is_conv, cur_is_tap))
                cur_is_conv, cur_is_tap = True, tap32 is not None
            cur = nxt
        assert self.layers[-1][0] == "conv" and cur_is_tap      # the deepest tensor is a tap without a consumer
        ctx.chain = True             # every gradient buffer is overwritten by its first writer: nothing to zero
        ctx.finalize()
        torch.cuda.synchronize()

    def forward(self, imgs: torch.Tensor, n0: int = 0, n1: Optional[int] = None, src0: Optional[int] = None,
                taps32: bool = True):
        """imgs: fp32 (2B,1,H,W): prediction images first, then targets.  Only images [n0, n1) are processed, so
        the target half can run early, beside other work, and the prediction half once it exists.  With src0 the
        images are read from imgs[src0 : src0 + n1 - n0] instead (a tensor that holds only one of the halves).
        taps32=False: no fp32 copies of the taps (bf16_targets)."""
        n1 = 2 * self.B if n1 is None else n1
        a0 = self.a0.images(n0, n1)
        src = imgs[n0:n1] if src0 is None else imgs[src0:src0 + n1 - n0]
        assert src.shape[0] == n1 - n0 and src.is_contiguous()
        ops.conv_c1_fwd(src, self.w0, a0.v, 1, bias=self.b0, in_scale=self.in_scale, in_shift=self.in_shift, act=1)
        for l in self.layers:
            if l[0] == "pool":
                p = l[1]
                ops.pool_fwd(p.xin.images(n0, n1).v, p.out.images(n0, n1).v, p.mode)
            else:
                _, core, bias, out, tap32 = l[:5]
                ops.conv2d(core.xin.images(n0, n1).v, core.wp, out.images(n0, n1).v, 3, 3, 1, 1, 1, 1, bias=bias, act=1,
                           y2=None if (tap32 is None or not taps32) else tap32[n0:n1])

    def forward_pred_fused(self, imgs: torch.Tensor, weight: float):
        """The prediction half (images [0, B)) with the perceptual MSE against the target half's fp32 taps (which
        forward(imgs, B, 2B) must have produced) fused into the tap layers: no fp32 copy of the prediction's
        features, no separate loss kernels.  Seeds each tap's gradient with weight * d loss / d tap; follow with
        loss_from_rows() and backward()."""
        B = self.B
        a0 = self.a0.images(0, B)
        ops.conv_c1_fwd(imgs[0:B], self.w0, a0.v, 1, bias=self.b0, in_scale=self.in_scale, in_shift=self.in_shift, act=1)
        for l in self.layers:
            if l[0] == "pool":
                p = l[1]
                ops.pool_fwd(p.xin.images(0, B).v, p.out.images(0, B).v, p.mode)
                continue
            _, core, bias, out, tap32 = l[:5]
            o = out.images(0, B)
            if tap32 is None:
                ops.conv2d(core.xin.images(0, B).v, core.wp, o.v, 3, 3, 1, 1, 1, 1, bias=bias, act=1)
            else:
                n = o.v.npix * o.v.C
                if self.bf16_targets:
                    ops.conv2d_tap_mse_bf16(core.xin.images(0, B).v, core.wp, bias, o.v, out.images(B, 2 * B).v, o.g,
                                            2.0 * weight / (n * len(self.taps)), core.mse_stats)
                else:
                    ops.conv2d_tap_mse(core.xin.images(0, B).v, core.wp, bias, o.v, tap32[B:], o.g,
                                       2.0 * weight / (n * len(self.taps)), core.mse_stats)

    def loss_from_rows(self, loss: torch.Tensor):
        """loss[0] += mean over taps of MSE(tap(pred), tap(target)) from the rows forward_pred_fused() left."""
        for l in self.layers:
            if l[0] == "conv" and l[4] is not None:
                core, v = l[1], l[3].images(0, self.B).v
                ops.stat_rows_total(core.mse_stats, core.mse_rows, v.C, 1, 1.0 / (v.npix * v.C * len(self.taps)), loss)

    def loss_and_seed(self, loss: torch.Tensor, weight: float):
        """loss[0] += mean over taps of MSE(tap(pred), tap(target)); seeds each tap's gradient (prediction
        half) with weight * d/d tap.  Gradient buffers must be zero."""
        B = self.B
        for t, t32 in self.taps:
            v = t.v
            n = B * v.H * v.W * v.C
            a, b = t32.data_ptr(), t32.data_ptr() + 4 * n
            g = t.g.images(0, B).ptr if self.ctx.grads else None
            ops.mse_tap(a, b, n, 1.0 / (n * len(self.taps)), loss, g, 2.0 * weight / (n * len(self.taps)),
                        accumulate=False)

    def mpl_loss_and_seed(self, loss: torch.Tensor, weight: float = 1.0):
        """MPL (src/loss.py:96-110): loss[0] += sum over taps of mean over (n, h, w) of ||tap(target) - tap(pred)||_2
        over channels / (h*w); seeds each tap's gradient (prediction half) with weight * d/d tap."""
        B = self.B
        for t, t32 in self.taps:
            v = t.v
            npix = B * v.H * v.W
            a, b = t32.data_ptr(), t32.data_ptr() + 4 * npix * v.C
            s = 1.0 / (npix * v.H * v.W)
            ops.l2norm_tap(a, b, npix, v.C, s, loss, t.g.images(0, B).ptr if self.ctx.grads else None, weight * s)

    def backward(self, dimg: torch.Tensor):
        """dimg: fp32 (B,1,H,W), += d loss / d prediction image."""
        B = self.B
        # Every gradient buffer holds the gradient w.r.t. the PRE-ReLU values: the ReLU backward is applied by
        # whatever writes the buffer (data-gradient epilogue or pooling backward).  The deepest tap has no
        # consumer, so its seed is masked here.
        o = self.layers[-1][3].images(0, B)
        ops.ew(ops.EW_ACTBWD, o.g, o.g, y=o.v, slope=0.0)
        for l in reversed(self.layers):
            if l[0] == "pool":
                p, seeded = l[1], l[2]
                xi = p.xin.images(0, B)
                ops.pool_bwd(xi.v, p.out.images(0, B).g, xi.g, p.mode, accumulate=seeded, relu_mask=True)
            else:
                _, core, bias, out, _t, in_is_conv, in_is_tap = l
                xi = core.xin.images(0, B)
                if in_is_conv:
                    core.dgrad(out.images(0, B).g, xi.g, accumulate=in_is_tap, mask=xi.v, mask_slope=0.0)
                else:
                    core.dgrad(out.images(0, B).g, xi.g, accumulate=False)
        ops.conv_c1_dgrad(self.a0.images(0, B).g, self.w0, dimg, scale=self.in_scale, accumulate=True)
