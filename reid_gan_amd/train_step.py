"""The fused G + D + VGG-perceptual + DSSIM training step and the inference + metrics pass.

TrainStep restates the hot loop of the reference's training script (src/train.py:153-177) as one fixed
sequence of kernel launches over preallocated HBM buffers, captured into a single HIP graph:

    pred = G(x)                                              train.py:153
    2 x { D(cat[y, pred]); DLoss; D backward; D Adam }       train.py:156-165
    gloss = CompoundLoss(pred, y, pred_neg of the last D forward)   train.py:172, loss.py:177-197
    G backward (through D as it was at that forward), G Adam        train.py:176-177

Work the script computes and then discards is not done: the D-loss gradient into G (pred is treated as
detached in the D iterations), VGG weight gradients, D weight gradients during the G backward.  The script's
own step raises on torch >= 1.5 (D is updated in place between the forward and the backward that uses it);
here the G gradient is taken before the last D Adam step, i.e. through the D that produced pred_neg.

Data parallel (one process per GPU, RCCL over xGMI): the two hinge means of DLoss are all-reduced before the
(non-linear) hinge so the result equals the single-GPU step at the same global batch; the discriminator's flat
gradient is averaged with one all-reduce per D iteration and the generator's in two buckets in the order its
backward pass completes them (bottleneck + decoder: 81 % of the parameters; then the encoder).  The last
discriminator gradient and the generator's first bucket are exchanged WHILE the second half of the generator
backward runs: each all-reduce is started the moment its gradients are complete and only waited for in front of the
Adam steps.  BatchNorm statistics stay per replica.

Look-ahead (lookahead=True, one rank): everything a step does that needs only the TARGET batch and weights that are
final before the step starts - the VGG features of the targets, the first discriminator iteration's spectral norms
and weight repacks, its pass over the target half of cat([y, pred]) - is done for the NEXT batch at the end of the
current step, on the stream of the last discriminator backward, beside the generator backward (some 270 small
dependent kernels that leave most of the chip idle).  step(x, y, y_next) takes the next batch's targets the way a
prefetching loader has them; a step whose y is not the y_next announced before (or the first one) prepares its own
targets first, so the results never depend on the announcement - only the speed does.
"""
from __future__ import annotations

from typing import Dict, Optional

import os

import torch

from . import ops
from .engine import Discriminator, Generator, State, Vgg, side_stream, N_LANES, VGG_LANE, DBWD_LANE, REPLICA_LANE0

F32 = torch.float32

# slots of the device scalar array
SC_MEAN_POS, SC_MEAN_NEG, SC_DLOSS, SC_CPOS, SC_CNEG, SC_VGG, SC_SSIM, SC_DSSIM, SC_GEN, SC_GLOSS = range(10)
SC_DLOSS1, SC_DLOSS2 = 10, 11

# Lifetime of captured graphs.  Destroying a many-stream captured graph has crashed this runtime twice: a host
# segmentation fault in hipGraphLaunch at the first replay of a capture made after such a graph had been destroyed,
# and - with graphs freed whenever Python's garbage collector dropped their owner - a host abort in the middle of
# unrelated CPU work some tests after several such owners had gone out of scope.  Neither sequence reproduces in a
# small script.  So, by default, a graph that has been captured is never destroyed: when its owner is closed
# (close(), or garbage collection) the graph object is parked in _RETAINED_GRAPHS until the process ends.  A parked
# graph costs its host-side node list and a private, near-empty memory pool (no step allocates during capture).
# RIED_FREE_GRAPHS=1 destroys graphs in close() instead - synchronise, drop in reverse capture order (segments
# chain their memory pools, the last captured must die first), synchronise - and never from the garbage collector.
_RETAINED_GRAPHS = []

# Stream-capture mode of every capture made here.  In the runtime's default ("global") mode a call that capture does
# not allow - an event query, a host allocation - made by ANY thread of the process while a capture is open fails in
# that thread and invalidates the capture.  RCCL's watchdog thread polls the events of finished collectives with
# exactly such queries: a capture begun right after the warm-up step's all-reduces was invalidated ("operation failed
# due to a previous error during capture") and the watchdog thread, whose query had failed, ended the process with
# std::terminate (SIGABRT on a secondary thread; native frames in libtorch_hip.so under a std::thread).  A loader's
# pinned-memory thread is another such caller.  "thread_local" confines the check to the capturing thread, which is
# the only one that launches into the captured streams.
CAPTURE_MODE = "thread_local"
# The perceptual loss compares the prediction's taps (fp32, before rounding) with the target's STORED bf16 activations
# instead of an fp32 copy of them: the target pass writes no fp32 taps (256 MB per step at batch 128) and the tap layers
# read half the bytes.  RIED_VGG_BF16_TARGETS=0: the fp32 copies.
VGG_BF16_TARGETS = os.environ.get("RIED_VGG_BF16_TARGETS", "1") != "0"
# The perceptual term (VGG16 on the prediction: forward with the fused tap MSE, then its data gradient) needs the
# generator's output and the target taps, nothing of the discriminator: it runs on a side stream from the end of the
# generator forward, beside the discriminator's first iteration and second forward, into a gradient image of its own
# that the generator-loss backward adds in.  Single-graph plans only (a side stream may not stay open across the
# collectives of a data-parallel step).  RIED_OVERLAP_VGG_PRED=0: in line, after the discriminator's second forward.
OVERLAP_VGG_PRED = os.environ.get("RIED_OVERLAP_VGG_PRED", "1") != "0"
VGGP_LANE = int(os.environ.get("RIED_VGGP_LANE", str(VGG_LANE)))      # (which side stream: they map to few hardware queues)


def _free_graphs() -> bool:
    return bool(os.environ.get("RIED_FREE_GRAPHS"))


def _retain(graph):
    """Called on every graph at creation: parks it unless RIED_FREE_GRAPHS is set."""
    if not _free_graphs():
        _RETAINED_GRAPHS.append(graph)
    return graph


def _park_on_del(owner):
    try:
        gs = getattr(owner, "graphs", None)
        gs = [g for g, _ in gs] if gs else []
        if getattr(owner, "graph", None) is not None:
            gs.append(owner.graph)
        for g in gs:
            if not any(g is r for r in _RETAINED_GRAPHS):
                _RETAINED_GRAPHS.append(g)
    except Exception:           # interpreter shutdown: globals may be gone, and then nothing needs keeping
        pass


def _dispose(graphs):
    """Let go of captured graphs (a list, oldest first).  Parked graphs stay alive; with RIED_FREE_GRAPHS=1 they are
    destroyed here, explicitly and in order, with the device idle on both sides."""
    graphs = [g for g in graphs if g is not None]
    if not graphs:
        return
    if not _free_graphs():
        for g in graphs:
            if not any(g is r for r in _RETAINED_GRAPHS):
                _RETAINED_GRAPHS.append(g)
        return
    torch.cuda.synchronize()
    while graphs:
        g = graphs.pop()          # last captured first
        g.reset()
        del g
    torch.cuda.synchronize()


class TrainStep:
    def __init__(self, g_sd: Dict[str, torch.Tensor], d_sd: Dict[str, torch.Tensor], vgg: Dict[str, torch.Tensor],
                 batch: int, H: int, W: int, lr: float = 1e-4, weights=(0.3, 0.5, 0.2), device="cuda",
                 use_graph: bool = True, process_group=None, lookahead: bool = False, fresh_d: bool = False,
                 sync_bn: bool = False):
        """fresh_d: the generator's adversarial term is taken from a NEW discriminator forward on the prediction after
        both discriminator updates - the `D_gen = Dnet(pred)` the reference's author left commented out at
        src/train.py:171 - instead of from the second iteration's forward (whose weights are one Adam step older).
        It costs one more discriminator forward on the prediction half and changes the reported gloss, so it is off
        by default.  sync_bn (with a process group): the generator's train-mode BatchNorm layers use the statistics of
        the GLOBAL batch (their per-channel sums are averaged over the ranks, forward and backward), so the step
        equals the single-process step at the global batch; per-replica statistics - what the nn.DataParallel left
        commented out at src/train.py:95-96 would compute - are the default.  One small all-reduce per BatchNorm
        layer and direction: the step then runs eagerly (no graph), for parity checks rather than throughput."""
        ops.require_gpu()
        self.B, self.H, self.W, self._lr = batch, H, W, float(lr)
        self.wv, self.ws, self.wg = weights
        self.device = device
        self.pg = process_group
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
        # With a process group the step is split into graph segments at the collectives and the all-reduces are
        # really issued, also for a group of one rank (which is how the path is exercised on a single GPU).
        self.segmented = process_group is not None
        self.gs, self.ds = State(g_sd, device), State(d_sd, device)
        B = batch
        self.xch = None
        if process_group is not None:
            from .dp import Exchanger
            self.xch = Exchanger(process_group)
        # The discriminator reads cat([target, prediction]) (train.py:158): the target batch and the generator's
        # output ARE the two halves of that buffer, so nothing is copied to build it.
        self.d_in = torch.zeros(2 * B, 1, H, W, dtype=F32, device=device)
        self.fresh_d = bool(fresh_d)
        self.sync_bn = bool(sync_bn) and process_group is not None
        if self.sync_bn:
            use_graph = False
        self.G = Generator(self.gs, B, H, W, grads=True, training=True, device=device,
                           grad_buckets=process_group is not None, out=self.d_in[B:],
                           bn_sync=self.xch.sync if self.sync_bn else None)
        self.D = Discriminator(self.ds, 2 * B, H, W, grads=True, training=True, device=device)
        self.V = Vgg(vgg, B, H, W, grads=True, device=device)
        self.V.bf16_targets = VGG_BF16_TARGETS
        self.x = torch.zeros(B, 1, H, W, dtype=F32, device=device)
        self.y = self.d_in[:B]
        self.dpred = torch.zeros(B, 1, H, W, dtype=F32, device=device)
        self.dpred_v = torch.zeros(B, 1, H, W, dtype=F32, device=device)      # the perceptual term's share (see OVERLAP_VGG_PRED)
        self._vggp_side = None
        self.overlap_vgg_pred = OVERLAP_VGG_PRED
        self.sc = torch.zeros(16, dtype=F32, device=device)
        self.ssim_sums = torch.zeros(B, 4, dtype=F32, device=device)
        self.gmap = torch.empty(3, B, H, W, dtype=F32, device=device)
        self.use_graph = use_graph
        import os
        self.overlap_vgg = not os.environ.get("RIED_NO_OVERLAP_VGG")
        self.overlap_gbwd = not os.environ.get("RIED_NO_OVERLAP_GBWD")
        # The discriminator's pass over the target half could run beside the generator forward; measured, the
        # two contend for the CUs and the step gets 0.12 ms slower, so it is off unless asked for.
        self.overlap_dreal = bool(os.environ.get("RIED_OVERLAP_DREAL"))
        self._d_side = None
        self.overlap_dprep = not os.environ.get("RIED_NO_OVERLAP_DPREP")
        self._d_prepared = False
        # look-ahead: see the module docstring.  With a process group the discriminator's Adam step waits for an
        # all-reduce that is only awaited after the generator backward, so there is nothing to run the work beside.
        self.lookahead = (bool(lookahead) and process_group is None and not self.fresh_d
                          and not os.environ.get("RIED_NO_LOOKAHEAD"))
        self.pre_dreal = self.lookahead and self.D.splittable and not os.environ.get("RIED_NO_PRE_DREAL")
        # which of the target-only pieces run at the end of the previous step (the others: beside the generator forward)
        self.la_vgg = self.lookahead and not os.environ.get("RIED_LA_VGG_FRONT")
        self.la_dreal = self.pre_dreal and not os.environ.get("RIED_LA_DREAL_FRONT")
        self.y_next = torch.zeros(B, 1, H, W, dtype=F32, device=device) if self.lookahead else None
        self._primed = False        # the targets in self.y have been prepared (by the previous step, or by _prime())
        self._d_ready = False       # the discriminator's weights are prepared for the next step's first iteration
        self._next_key = None       # identifies the tensor announced as y_next
        self._next_ref = None       # ... and keeps it alive, so that no other tensor can take its address meanwhile
        self._ynext_is_y = False    # the y_next buffer holds a copy of self.y (nothing was announced)
        self._warmed = False        # every kernel of the step has been loaded (an eager step has run)
        self.graphs = None          # [(graph, tensor to all-reduce after it or None)]
        self.steps_done = 0
        self.segment = -1
        self._inflight = {}

    @property
    def lr(self) -> float:
        """Generator learning rate (the discriminator steps at four times it: train.py:104-105).  The rate is an
        argument of the captured Adam launches, so assigning a different value drops the captured graphs and the
        next step() captures again."""
        return self._lr

    @lr.setter
    def lr(self, value: float):
        value = float(value)
        if value != self._lr:
            self._lr = value
            self._drop_graphs()

    def _drop_graphs(self):
        if self.graphs is not None:
            graphs, self.graphs = [g for g, _ in self.graphs], None
            _dispose(graphs)

    def close(self):
        """Release the captured graphs (see _dispose); the plan can still step afterwards (it captures again).
        Call it when done with the object instead of leaving that to the garbage collector."""
        self._drop_graphs()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):          # never destroy a graph from the garbage collector: park what close() did not see
        if _park_on_del is not None:          # (None once the interpreter is tearing the module down)
            _park_on_del(self)

    @property
    def exchange(self):
        """(collectives started so far, in flight now): where a stuck rank stands (read by watchdogs)."""
        return (self.xch.issued if self.xch is not None else 0, len(self._inflight))

    # ------------------------------------------------------------------ collectives
    # The step (_body) is a generator of exchange points; between two of them it only launches kernels.  Each
    # point is a list of events handled in order:
    #   ("sync", t)         all-reduce t now, the next kernels need the result
    #   ("issue", key, t)   t is complete on the current stream: start its all-reduce
    #   ("wait", key)       the next kernels need the result of that all-reduce
    def _handle(self, events):
        if self.xch is None or events is None:
            return
        for ev in events:
            if ev[0] == "sync":
                self.xch.sync(ev[1])
            elif ev[0] == "issue":
                self._inflight[ev[1]] = self.xch.issue(ev[2])
            else:
                self.xch.wait(self._inflight.pop(ev[1]))

    # ------------------------------------------------------------------ the step
    # With one rank the events are skipped and the whole step is one HIP graph; with several ranks each stretch
    # between two events is its own graph and the RCCL all-reduces are started / awaited between the replays, so
    # nothing synchronises with the host either way and a started all-reduce overlaps the graphs replayed after it.
    # Two whole-network overlaps on side streams (graph capture records them as parallel paths):
    #  * the VGG features of the TARGET images depend on nothing the step computes, so they run beside the
    #    generator forward, whose small layers leave most of the chip idle;
    #  * the generator backward (small, latency-bound kernels) runs beside the last discriminator backward and
    #    Adam step (large kernels); both only need what the generator-loss backward has produced by then.
    @staticmethod
    def _fork(lane: int, enabled: bool = True):
        if not enabled:
            return torch.cuda.current_stream()
        s = side_stream(lane)
        s.wait_stream(torch.cuda.current_stream())
        return s

    @staticmethod
    def _join(s):
        if s != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_stream(s)

    def _d_iteration(self, last: bool, first: Optional[bool] = None):
        """One discriminator iteration (train.py:156-165).  first: it is the step's first one (what was prepared
        beside the generator forward belongs to it); last: the generator loss and backward hang on this iteration's
        forward (False for both iterations of a fresh_d step)."""
        B, sc, D = self.B, self.sc, self.D
        first = (not last) if first is None else first
        if first and self.pre_dreal:
            out = D.forward_images(self.d_in, B, 2 * B)      # the target half ran at the end of the previous step
        elif first and self._d_side is None and self._d_prepared:
            out = D.forward_images(self.d_in, 0, 2 * B)      # spectral norms and repacks ran beside the generator forward
        elif not first or self._d_side is None:
            out = D.forward(self.d_in)
        else:          # the target half has been running beside the generator forward (see _body)
            out = D.forward_images(self.d_in, B, 2 * B)
            self._join(self._d_side)
            self._d_side = None
        n = out.v.npix // 2 * out.v.C                    # elements per half = B * 8192 at 64x64
        sc[SC_MEAN_POS:SC_MEAN_NEG + 1].zero_()
        ops.dot(out.v.images(0, B), None, 1.0 / n, sc[SC_MEAN_POS:])
        ops.dot(out.v.images(B, 2 * B), None, 1.0 / n, sc[SC_MEAN_NEG:])
        yield [("sync", sc[SC_MEAN_POS:SC_MEAN_NEG + 1])]    # the hinge acts on the GLOBAL batch means
        ops.dloss_scalars(sc, n, keep=sc[SC_DLOSS1 if first else SC_DLOSS2:])
        if last:       # the generator's gradient goes through D as it is now; D's own backward reuses the buffers after
            self._g_loss_backward(n)
        # D's own backward (large kernels) on a side stream beside the generator backward (small, latency-bound
        # kernels; its Inception branches fork from the stream it runs on).  Data parallel: the side stream rejoins
        # when the generator's first gradient bucket is complete, and both exchanges start there, in flight during
        # the second half of the generator backward.
        side = self._fork(DBWD_LANE, last and self.overlap_gbwd)
        with torch.cuda.stream(side):
            D.ctx.zero_backward()
            self.ds.zero_grad()
            ops.fill(out.g.images(0, B), 1.0, s=sc[SC_CPOS:])
            ops.fill(out.g.images(B, 2 * B), 1.0, s=sc[SC_CNEG:])
            D.backward(0, 2 * B, wgrad=True)
            if last and self.lookahead:      # D is final for this step: the next batch's targets, beside the G backward
                self.ds.adam(4.0 * self.lr)
                self.y.copy_(self.y_next)
                self._prep_targets()
        if not last:
            yield [("sync", self.ds.flat_grad)]
            self.ds.adam(4.0 * self.lr)
            return
        yield from self._g_backward(side)
        if self.lookahead:
            self._join(side)
        else:
            self.ds.adam(4.0 * self.lr)

    def _g_backward(self, side=None):
        """The generator backward from self.dpred, bucket by bucket.  side: the stream the last discriminator backward
        runs on (its flat gradient is exchanged together with the generator's first bucket), or None when the
        discriminator has been updated already."""
        self.gs.zero_grad()
        gr, first = self.G.bucket_ranges, True
        for b in self.G.backward_buckets(self.dpred):
            ev = [("issue", ("g", b), self.gs.flat_grad[gr[b][0]:gr[b][1]])]
            if first and side is not None:
                self._join(side)
                ev.insert(0, ("issue", "d", self.ds.flat_grad))
            first = False
            if b == 0:      # the last bucket: everything is needed next
                ev += ([("wait", "d")] if side is not None else []) + [("wait", ("g", k)) for k in reversed(range(len(gr)))]
            yield ev

    def _fresh_d_g_loss(self):
        """fresh_d: D(pred) with the discriminator as it is after both updates (a training-mode forward: one more
        power iteration of every spectral norm, as the module's forward hook would run), then the generator loss and
        backward through it."""
        B, sc, D = self.B, self.sc, self.D
        if D.splittable:
            D.prepare()
            out = D.forward_images(self.d_in, B, 2 * B)
        else:
            out = D.forward(self.d_in)
        n = out.v.npix // 2 * out.v.C
        sc[SC_MEAN_NEG:SC_MEAN_NEG + 1].zero_()
        ops.dot(out.v.images(B, 2 * B), None, 1.0 / n, sc[SC_MEAN_NEG:])
        yield [("sync", sc[SC_MEAN_NEG:SC_MEAN_NEG + 1])]
        self._g_loss_backward(n)
        yield from self._g_backward(None)

    def _prep_targets(self, weights: bool = True):
        """What the step needs from the target batch in self.y and the weights as they are before it starts.
        weights=False: the discriminator's weight preparation (it advances the spectral norms' power iteration,
        so it must run exactly once per discriminator update) has been done already; only the targets are new."""
        B = self.B
        if weights:
            self.D.prepare()
        if self.la_vgg:
            self.V.forward(self.y, B, 2 * B, src0=0, taps32=not VGG_BF16_TARGETS)
        if self.la_dreal:
            self.D.forward_images(self.d_in, 0, B)

    def _g_loss_backward(self, n: int):
        B, H, W, sc = self.B, self.H, self.W, self.sc
        pred, y = self.G.out, self.y
        self.dpred.zero_()
        sc[SC_SSIM:SC_SSIM + 1].zero_()
        # perceptual term (loss.py:179-189); the target half of the features was started in _body()
        if self._vggp_side is None:          # (else it has been running since the generator forward: joined below)
            sc[SC_VGG:SC_VGG + 1].zero_()
            if self._vgg_side is not None:
                self._join(self._vgg_side)
                self._vgg_side = None
            self.dpred_v.zero_()             # (its own image here too: both forms add the three terms in one order)
            self._vgg_pred_pass(self.dpred_v)
        # DSSIM term on raw HU (loss.py:191): ssim(target, pred), gradient to pred
        self.ssim_sums.zero_()
        ops.ssim_fwd(pred, None, y, 0.01 ** 2, 0.03 ** 2, self.ssim_sums, gmap=self.gmap)
        ops.sum_f32(self.ssim_sums, B, 4, 1.0 / (B * H * W), sc[SC_SSIM:])
        ops.ssim_bwd(pred, y, self.gmap, -self.ws * 0.5 / (B * H * W), self.dpred, accumulate=True)
        # generator term 1 - mean(pred_neg) (loss.py:154-155), through D as it is now
        out = self.D.out
        self.D.ctx.zero_backward()
        ops.fill(out.g.images(B, 2 * B), -self.wg / n)
        self.D.backward(B, 2 * B, wgrad=False, dimg=self.dpred)
        if self._vggp_side is not None:
            self._join(self._vggp_side)
            self._vggp_side = None
        ops.axpy_f32(self.dpred, self.dpred_v, 1.0)
        ops.gloss_scalars(sc, self.wv, self.ws, self.wg)

    def _vgg_pred_pass(self, dimg: torch.Tensor):
        self.V.ctx.zero_backward()
        self.V.forward_pred_fused(self.G.out, self.wv)      # the MSE against the target taps rides in the tap layers
        self.V.loss_from_rows(self.sc[SC_VGG:])
        self.V.backward(dimg)

    def _start_vgg_pred(self):
        """After the generator forward: the perceptual term on the VGG side stream (see OVERLAP_VGG_PRED)."""
        self._vggp_side = None
        if not (self.overlap_vgg_pred and not self.segmented):
            return
        self.sc[SC_VGG:SC_VGG + 1].zero_()
        self.dpred_v.zero_()
        s = self._vgg_side if self._vgg_side is not None else side_stream(VGGP_LANE)
        s.wait_stream(torch.cuda.current_stream())      # the prediction, the two clears (and, in stream order, the target taps)
        with torch.cuda.stream(s):
            self._vgg_pred_pass(self.dpred_v)
        self._vgg_side, self._vggp_side = None, s
        if os.environ.get("RIED_VGGP_JOIN_NOW"):       # (diagnostic: the side stream, but nothing beside it)
            self._join(s)

    def _body(self):
        B = self.B
        if self.lookahead:         # the targets were prepared by the previous step (or _prime())
            self._vgg_side, self._d_side, self._d_prepared = None, None, True
            if not self.la_vgg:
                self._vgg_side = self._fork(VGG_LANE, self.overlap_vgg)
                with torch.cuda.stream(self._vgg_side):
                    self.V.forward(self.y, B, 2 * B, src0=0, taps32=not VGG_BF16_TARGETS)
            side = None
            if self.pre_dreal and not self.la_dreal:
                side = self._fork(DBWD_LANE)
                with torch.cuda.stream(side):
                    self.D.forward_images(self.d_in, 0, B)
            pred = self.G.forward(self.x)
            if side is not None:
                self._join(side)
            assert pred.data_ptr() == self.d_in[B:].data_ptr()
            self._start_vgg_pred()
            yield from self._d_iteration(last=False)
            yield from self._d_iteration(last=True)
            self.gs.adam(self.lr)
            return
        self._vgg_side = self._fork(VGG_LANE, self.overlap_vgg)      # target features beside the generator forward
        # ... and so do the first discriminator iteration's spectral norms and weight repacks (small kernels that need
        # only D's weights): 0.1 ms off the critical path
        dreal = self.overlap_dreal and self.D.splittable      # (then the preparation runs with that pass, once)
        self._d_prepared = self.overlap_vgg and self.overlap_dprep and not dreal
        with torch.cuda.stream(self._vgg_side):
            if self._d_prepared:
                self.D.prepare()
            self.V.forward(self.y, B, 2 * B, src0=0, taps32=not VGG_BF16_TARGETS)
        # The first discriminator iteration's weight preparation (spectral norms, packing) and its forward pass on
        # the TARGET images need nothing the generator computes: they run beside the generator forward too, whose
        # small dependent kernels leave most of the chip idle.
        self._d_side = None
        if dreal:
            self._d_side = self._fork(DBWD_LANE)
            with torch.cuda.stream(self._d_side):
                self.D.prepare()
                self.D.forward_images(self.d_in, 0, B)
        pred = self.G.forward(self.x)
        # (a side stream may not stay open across a collective = graph segment boundary; and the discriminator
        # forward that comes next reads the weights repacked on it)
        if self.segmented or self._d_prepared:
            self._join(self._vgg_side)
            self._vgg_side = None
        assert pred.data_ptr() == self.d_in[B:].data_ptr()
        self._start_vgg_pred()
        yield from self._d_iteration(last=False)
        if self.fresh_d:
            yield from self._d_iteration(last=False, first=False)
            yield from self._fresh_d_g_loss()
        else:
            yield from self._d_iteration(last=True)
        self.gs.adam(self.lr)

    def _run_eager(self):
        for ev in self._body():
            self._handle(ev)

    def weights_changed(self, d_prepared: bool = False):
        """Tell the plan that the networks' weights were written from outside (a checkpoint restore): what a
        look-ahead plan prepared for its next step with the old weights is discarded and done again.
        d_prepared: the restored spectral-norm vectors already include the next step's power iteration (a checkpoint
        written after a look-ahead step: that iteration belongs to the next step and had been run when the file was
        saved).  A look-ahead plan then rebuilds sigma and the packed weights from those vectors WITHOUT iterating
        again, so the resumed run continues exactly as the uninterrupted one; a plain plan cannot take the iteration
        back and runs one more than the uninterrupted plain run would have."""
        self._primed = self._d_ready = False
        self._resume_prepared = bool(d_prepared) and self.lookahead

    def snapshot_discriminator(self):
        """A device copy of everything the discriminator's training owns (weights, spectral-norm vectors, Adam
        moments and step count), for reseed_discriminator()."""
        st = self.ds
        return {"flat": st.flat.clone(), "b": {k: v.clone() for k, v in st.b.items()},
                "adam": (st.adam_m.clone(), st.adam_v.clone(), st.adam_step.clone())}

    def reseed_discriminator(self, snap):
        """Put the discriminator back to a snapshot between two steps (device copies, no host synchronisation).
        DLoss is a hinge on the batch MEANS (src/loss.py:141-142): once the discriminator separates the batch both
        gates close and its gradient is exactly zero until the generator catches up.  A benchmark that wants the
        gates open - gradients with real values in every discriminator kernel - re-seeds the discriminator every few
        steps.  A look-ahead plan redoes what it had prepared from the old weights (spectral norms, packed weights
        and the pass over the next batch's targets; the VGG target features do not depend on the discriminator)."""
        st = self.ds
        st.flat.copy_(snap["flat"])
        for k, v in snap["b"].items():
            st.b[k].copy_(v)
        for dst, src in zip((st.adam_m, st.adam_v, st.adam_step), snap["adam"]):
            dst.copy_(src)
        if self.lookahead and self._primed:
            self.D.prepare()
            if self.la_dreal:
                self.D.forward_images(self.d_in, 0, self.B)
            self._d_ready = True
        else:
            self._d_ready = False

    @staticmethod
    def _key(t: torch.Tensor):
        return (t.data_ptr(), t._version, tuple(t.shape))

    def _prime(self):
        """Look-ahead: prepare the targets in self.y now (first step, or a y that was not announced)."""
        if getattr(self, "_resume_prepared", False):      # vectors restored one iteration ahead: sigma + repack only
            self.D.prepare(power_iter=False)
            self._d_ready, self._resume_prepared = True, False
        self._prep_targets(weights=not self._d_ready)
        self._primed = self._d_ready = True

    def step(self, x: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
             y_next: Optional[torch.Tensor] = None):
        """One training iteration on (x, y), fp32 (B,1,H,W) device tensors.  No host synchronisation; read
        the losses afterwards with losses().  y_next (look-ahead plans only): the targets of the batch the NEXT
        call will be given; that call finds its targets prepared if its y is this very tensor, unmodified, and
        prepares them itself otherwise.  Without arguments the buffers self.x / self.y are used as they are (a
        look-ahead plan then repeats the batch unless y_next announces another)."""
        if x is not None:
            self.x.copy_(x, non_blocking=True)
            # (the announced tensor is kept alive in _next_ref, so its address cannot have gone to another batch)
            announced = (self.lookahead and self._primed and self._next_ref is not None
                         and self._key(y) == self._next_key)
            if not announced:            # (else self.y holds these targets already, prepared)
                self.y.copy_(y, non_blocking=True)
                self._primed = False
        if self.lookahead:
            if y_next is not None:
                self.y_next.copy_(y_next, non_blocking=True)
                self._next_key, self._next_ref, self._ynext_is_y = self._key(y_next), y_next, False
            elif x is not None or not self._ynext_is_y:      # nothing announced: the end of the step re-prepares this batch
                self.y_next.copy_(self.y)
                self._next_key, self._next_ref, self._ynext_is_y = None, None, True
            if self.use_graph and self.graphs is None:
                self._capture()
            if not self._primed:
                self._prime()
        if not self.use_graph:
            self._run_eager()
            self._warmed = True
        else:
            if self.graphs is None:
                self._capture()
            for i, (g, ev) in enumerate(self.graphs):
                self.segment = i          # (where a stuck rank stands: read by watchdogs)
                g.replay()
                self._handle(ev)
        self.steps_done += 1           # (look-ahead: self.y now holds the y_next buffer's batch, prepared)

    def _capture(self):
        # one eager step first (loads every kernel, warms RCCL), then put the state back.  (Look-ahead: that step
        # leaves targets prepared with weights that are then taken back, so the caller primes again; a re-capture -
        # a new learning rate - of a plan whose kernels are loaded skips it and keeps what is prepared.)
        if not (self.lookahead and self._warmed):
            snap = self._snapshot()
            if self.lookahead:
                snap["y"] = self.y.clone()       # (the step ends by moving the y_next buffer into self.y)
                self._prime()
            self._run_eager()
            torch.cuda.synchronize()
            self._restore(snap)
            self._primed, self._d_ready, self._warmed = False, False, True
        graphs = []
        gen = self._body()
        pool = None
        done = False
        while not done:
            g = _retain(torch.cuda.CUDAGraph())
            with torch.cuda.graph(g, pool=pool, capture_error_mode=CAPTURE_MODE):
                t = None
                while True:
                    try:
                        t = next(gen)
                    except StopIteration:
                        done, t = True, None
                        break
                    if self.segmented:
                        break                 # close this graph at the collective
            pool = g.pool()
            graphs.append((g, t))
        self.graphs = graphs
        # the capture pass did not execute anything: state is still the restored one

    def _snapshot(self):
        s = {"g": self.gs.flat.clone(), "d": self.ds.flat.clone(),
             "gb": {k: v.clone() for k, v in self.gs.b.items()}, "db": {k: v.clone() for k, v in self.ds.b.items()},
             "adam": [(st.adam_m.clone(), st.adam_v.clone(), st.adam_step.clone()) for st in (self.gs, self.ds)]}
        return s

    def _restore(self, s):
        self.gs.flat.copy_(s["g"]); self.ds.flat.copy_(s["d"])
        for k, v in s["gb"].items():
            self.gs.b[k].copy_(v)
        for k, v in s["db"].items():
            self.ds.b[k].copy_(v)
        for st, (m, v, n) in zip((self.gs, self.ds), s["adam"]):      # e.g. moments restored from a checkpoint
            st.adam_m.copy_(m); st.adam_v.copy_(v); st.adam_step.copy_(n)
        if "y" in s:
            self.y.copy_(s["y"])

    def losses(self) -> Dict[str, float]:
        """Host read of the last step's scalars (synchronises).  Raises if a one-launch BatchNorm backward
        (RIED_BN_BWD_ONE=1) ever gave up at its grid barrier: the gradients of that launch are wrong."""
        from . import engine
        if engine.BN_BWD_ONE and ops.barrier_timeouts():
            raise RuntimeError(f"{ops.barrier_timeouts()} blocks timed out at a grid barrier (RIED_BN_BWD_ONE): the "
                               "launch was not resident at once; BatchNorm gradients of this run are invalid")
        s = self.sc.clone()
        if self.xch is not None:
            self.xch.sync(s)
        s = s.cpu()
        return {"dloss1": s[SC_DLOSS1].item(), "dloss2": s[SC_DLOSS2].item(), "gloss": s[SC_GLOSS].item(),
                "vgg": s[SC_VGG].item(), "dssim": s[SC_DSSIM].item(), "gen": s[SC_GEN].item(),
                "ssim": s[SC_SSIM].item()}


class InferenceEngine:
    """The evaluation loop body of the reference (src/test.py:125-144) for a batch of slices: generator forward
    (train-mode BatchNorm, as the script never calls eval()), clamp to [lo, hi], and the six numbers of
    measure.compute_measure per slice, from one read of x, y and pred."""

    def __init__(self, g_sd: Dict[str, torch.Tensor], batch: int, H: int, W: int, device="cuda", train_mode_bn=True,
                 lo=-160.0, hi=240.0, use_graph=True, per_slice_stats=False):
        """per_slice_stats: BatchNorm statistics of every slice on its own, as in the reference's loop over a
        batch-1 loader (src/test.py:107-131), inside the batched kernels (H, W divisible by 256).  Default: the
        generator applied to the batch (statistics over the batch)."""
        ops.require_gpu()
        self.gs = State(g_sd, device)
        self.G = Generator(self.gs, batch, H, W, grads=False, training=train_mode_bn, device=device,
                           per_image_stats=per_slice_stats)
        self.B, self.H, self.W, self.lo, self.hi = batch, H, W, lo, hi
        self.sums = torch.zeros(batch, 4, dtype=F32, device=device)
        self.x = torch.zeros(batch, 1, H, W, dtype=F32, device=device)
        self.y = torch.zeros(batch, 1, H, W, dtype=F32, device=device)
        self.use_graph, self.graph = use_graph, None
        self.refresh_weights()

    def refresh_weights(self, g_sd: Optional[Dict[str, torch.Tensor]] = None):
        """The weights of an evaluation loop do not change between batches, so their packed bf16 copies are built here,
        once, and not at the head of every pass (a launch over every weight of the network).  Call again - with a state
        dict to load, or right after writing into self.gs - when they do change (until then the layers that read the
        fp32 weights directly see the new values and the others the old)."""
        for st, plan in self._plans():
            if g_sd is not None:
                st.load_state_dict(g_sd)
            plan.ctx.pack_all()
            plan.ctx.repack = bool(os.environ.get("RIED_INFER_REPACK"))      # (A/B switch: pack at every pass again)

    def _plans(self):
        return [(self.gs, self.G)]

    def _body(self):
        pred = self.G.forward(self.x)
        dr = self.hi - self.lo
        self.sums.zero_()
        ops.ssim_fwd(self.x, pred, self.y, (0.01 * dr) ** 2, (0.03 * dr) ** 2, self.sums, clamp=(self.lo, self.hi))

    def run(self, x: torch.Tensor, y: torch.Tensor):
        """Returns (pred, sums) on the device; sums[n] = [sum ssim(x,y), sum (x-y)^2, sum ssim(pred,y), sum (pred-y)^2]
        over the clamped images.  The whole pass is one HIP graph replay (train-mode BatchNorm updates the running
        statistics on every call, as the reference's evaluation loop does)."""
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        if not self.use_graph:
            self._body()
        else:
            if self.graph is None:
                snap = {k: v.clone() for k, v in self.gs.b.items()}
                self._body()                       # loads every kernel before the capture
                torch.cuda.synchronize()
                for k, v in snap.items():
                    self.gs.b[k].copy_(v)
                self.graph = _retain(torch.cuda.CUDAGraph())
                with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
                    self._body()
            self.graph.replay()
        return self.G.out, self.sums

    def close(self):
        """Release the captured graph (see _dispose); a later run() captures again."""
        if self.graph is not None:
            g, self.graph = self.graph, None
            _dispose([g])

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):          # never destroy a graph from the garbage collector: park what close() did not see
        if _park_on_del is not None:          # (None once the interpreter is tearing the module down)
            _park_on_del(self)

    def measures(self, sums: torch.Tensor):
        """Per-slice ((psnr, ssim, rmse) original, (psnr, ssim, rmse) prediction) as Python floats."""
        import math
        s = sums.double().cpu() / (self.H * self.W)
        dr = self.hi - self.lo
        out = []
        for i in range(s.shape[0]):
            o = (10 * math.log10(dr * dr / s[i, 1].item()), s[i, 0].item(), math.sqrt(s[i, 1].item()))
            p = (10 * math.log10(dr * dr / s[i, 3].item()), s[i, 2].item(), math.sqrt(s[i, 3].item()))
            out.append((o, p))
        return out


class SliceInference(InferenceEngine):
    """The reference's evaluation loop as it is written - a batch-1 loader, so train-mode BatchNorm normalises
    every slice with that slice's own statistics (src/test.py:107-131) - at batch throughput.  Each slice of the
    batch goes through its own batch-1 generator plan; `lanes` plans run side by side on parallel streams (a
    single batch-1 pass is some 360 short dependent launches that leave most of the chip idle), all inside one
    HIP graph, and one metrics launch covers the batch.  InferenceEngine, in contrast, is the generator applied
    to the whole batch at once (statistics over the batch).  Every plan keeps its own copy of the BatchNorm
    running statistics (the loop's outputs do not depend on them)."""

    def __init__(self, g_sd: Dict[str, torch.Tensor], batch: int, H: int, W: int, device="cuda", lanes: int = 16,
                 lo=-160.0, hi=240.0, use_graph=True):
        from . import engine
        ops.require_gpu()
        self.B, self.H, self.W, self.lo, self.hi = batch, H, W, lo, hi
        self.lanes = max(1, min(lanes, batch))
        self.sums = torch.zeros(batch, 4, dtype=F32, device=device)
        self.x = torch.zeros(batch, 1, H, W, dtype=F32, device=device)
        self.y = torch.zeros(batch, 1, H, W, dtype=F32, device=device)
        self.pred = torch.empty(batch, 1, H, W, dtype=F32, device=device)
        self.states = [State(g_sd, device) for _ in range(batch)]
        self.gs = self.states[0]
        self.plans = [Generator(self.states[i], 1, H, W, grads=False, training=True, device=device,
                                out=self.pred[i:i + 1]) for i in range(batch)]
        self.G = self.plans[0]
        self._engine = engine
        self.use_graph, self.graph = use_graph, None
        self.refresh_weights()

    def _plans(self):
        return list(zip(self.states, self.plans))

    def _body(self):
        eng = self._engine
        main = torch.cuda.current_stream()
        saved, eng.PAR_BRANCHES = eng.PAR_BRANCHES, False      # the plans, not their Inception branches, run in parallel
        try:
            for lane in range(self.lanes):
                s = side_stream(REPLICA_LANE0 + lane)
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    for i in range(lane, self.B, self.lanes):
                        self.plans[i].forward(self.x[i:i + 1])
            for lane in range(self.lanes):
                main.wait_stream(side_stream(REPLICA_LANE0 + lane))
        finally:
            eng.PAR_BRANCHES = saved
        dr = self.hi - self.lo
        self.sums.zero_()
        ops.ssim_fwd(self.x, self.pred, self.y, (0.01 * dr) ** 2, (0.03 * dr) ** 2, self.sums, clamp=(self.lo, self.hi))

    def run(self, x: torch.Tensor, y: torch.Tensor):
        """Returns (pred, sums) as InferenceEngine.run does."""
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)
        if not self.use_graph:
            self._body()
        else:
            if self.graph is None:
                self._body()                       # loads every kernel before the capture
                torch.cuda.synchronize()
                self.graph = _retain(torch.cuda.CUDAGraph())
                with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
                    self._body()
            self.graph.replay()
        return self.pred, self.sums
