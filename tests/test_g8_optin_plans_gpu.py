"""Plans that are OFF by default (measured slower or equal, kept switchable), the look-ahead variants and the fresh_d
step: each must compute the same step as the default plan.  They sit in the last GPU test file on purpose: a
failure here cannot hide the default path's parity evidence, which the earlier files hold."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ried_oracle as O
from reid_gan_amd import engine, ops
from reid_gan_amd.synth import synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep, InferenceEngine
from gpu_util import DEV, emu, act_close, grad_close, g_state, d_state, weights_agree, _walk


def test_optional_step_overlaps_give_the_same_step(monkeypatch):
    """The switchable parts of the captured step - the discriminator's weight preparation beside the generator
    forward (on by default) and its pass over the target half there too (off by default: measured slower) - change
    the order of launches, not the arithmetic: same losses and weights as the step without them."""
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=98)
    monkeypatch.setattr(engine, "BN_FOLD", False)        # fixed-order reductions: the runs must agree closely
    plain = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    plain.overlap_dprep = False
    dreal = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    dreal.overlap_dreal = True
    default = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    assert default.overlap_dprep and not default.overlap_dreal
    for ts in (plain, dreal, default):
        ts.step(x.to(DEV), y.to(DEV))
        ts.step()
    torch.cuda.synchronize()
    a = plain.losses()
    for other in (dreal, default):
        b = other.losses()
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-6, (k, a[k], b[k])
        weights_agree(plain, other)


@pytest.mark.parametrize("use_graph", [True, False])
def test_lookahead_steps_equal_plain_steps(monkeypatch, use_graph):
    """lookahead=True: the target-only work of a step (VGG features of y, the first discriminator iteration's
    weight preparation and its pass over the target half) is done for the NEXT batch at the end of the current
    step.  The same kernels on the same data in another place: three steps over three different batches give the
    losses and weights of the plain plan - with every next batch announced, with none announced (each step then
    prepares its own targets), and with a wrong announcement (detected: the tensor passed is not the announced one)."""
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    batches = [tuple(t.to(DEV) for t in synth_batch(B, H, seed=60 + i)) for i in range(4)]
    monkeypatch.setattr(engine, "BN_FOLD", False)        # fixed-order reductions: the runs must agree closely
    plain = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=use_graph)
    ahead = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=use_graph, lookahead=True)
    blind = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=use_graph, lookahead=True)
    wrong = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=use_graph, lookahead=True)
    assert ahead.lookahead and ahead.pre_dreal and not plain.lookahead
    for i in range(3):
        x, y = batches[i]
        plain.step(x, y)
        ahead.step(x, y, y_next=batches[i + 1][1])
        blind.step(x, y)
        wrong.step(x, y, y_next=batches[(i + 2) % 4][1])
        torch.cuda.synchronize()
        a = plain.losses()
        for name, other in (("ahead", ahead), ("blind", blind), ("wrong", wrong)):
            b = other.losses()
            for k in a:
                assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-6, (name, i, k, a[k], b[k])
    for other in (ahead, blind, wrong):
        weights_agree(plain, other, steps=3)
    # replays without arguments repeat the batch: same as passing it again
    x, y = batches[2]
    plain.step(x, y); plain.step(x, y)
    blind.step(); blind.step()
    torch.cuda.synchronize()
    a, b = plain.losses(), blind.losses()
    for k in a:
        assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-6, (k, a[k], b[k])


def test_training_plan_with_the_one_launch_batchnorm_backward(monkeypatch):
    """RIED_BN_BWD_ONE=1: the BatchNorm backward of small layers as one launch with a grid barrier (off by default:
    measured equal).  The same folded sums in another order: the step agrees with the default plan up to the rounding
    of those sums, the launch count drops, and no block ever gave up at a barrier."""
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=96)
    base = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    monkeypatch.setattr(engine, "BN_BWD_ONE", True)
    one = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    n_one = sum(1 for m in one.G.body.layers for l in _walk(m) if isinstance(l, engine.ConvBN) and l.bwd_one)
    assert n_one >= 40, n_one
    assert not any(isinstance(l, engine.ConvBN) and l.bwd_one for m in base.G.body.layers for l in _walk(m))
    for ts in (base, one):
        ts.step(x.to(DEV), y.to(DEV))
    torch.cuda.synchronize()
    a, b = base.losses(), one.losses()
    for k in ("dloss1", "dloss2", "gloss", "vgg", "dssim", "gen"):
        assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-5, (k, a[k], b[k])
    gf = torch.cat([one.gs.g[k].float().flatten() for k in sorted(one.gs.g)]).cpu()
    gb = torch.cat([base.gs.g[k].float().flatten() for k in sorted(base.gs.g)]).cpu()
    grad_close(gf, gb, "all generator gradients", cos=0.98, ratio=0.05)
    assert ops.barrier_timeouts() == 0


def test_inference_with_the_fused_input_transform(monkeypatch):
    """Forward-only plans can let the consumer of an inner BasicConv2d layer normalise that layer's raw output inside
    its own kernel (ried_conv2d_xf).  By default only stride-2 consumers do (the others were measured slower);
    here: no consumer, the default, and every consumer.  Same prediction and metrics as the plan that writes every
    normalised tensor (the transform is the apply pass's arithmetic, so the tolerance only covers reduction order)."""
    sd = g_state()
    B, H = 4, 256
    x, y = synth_batch(B, H, seed=79)
    xd, yd = x.to(DEV), y.to(DEV)

    def run(mode):
        monkeypatch.setattr(engine, "XF_FUSE_MODE", mode)
        monkeypatch.setattr(engine, "XF_FUSE", mode not in ("", "0"))
        e = InferenceEngine(sd, B, H, H)
        n = len([l for m in e.G.body.layers for l in _walk(m)
                 if isinstance(l, engine.ConvBN) and l.core.xf_src is not None])
        # deconv2 normalises deconv1's raw output (its slice of the concatenation) inside the deconvolution halo kernel
        assert (getattr(e.G.last.core, "xf_mixed", None) is not None) == (mode != "0")
        p, s_ = e.run(xd, yd)
        return n, p.cpu().clone(), e.measures(s_)

    n0, pa, ma = run("0")
    n2, p2, m2 = run("s2")
    n1, pb, mb = run("1")
    assert n0 == 0 and 3 <= n2 < n1 and n1 >= 10, (n0, n2, n1)
    for p, m, what in ((p2, m2, "stride-2 consumers"), (pb, mb, "every consumer")):
        act_close(p, pa, f"fused ({what}) vs separate normalisation", mx=2e-2, mean=3e-3)
        for i in range(B):
            for u, v in zip(m[i][1], ma[i][1]):
                assert abs(u - v) <= 1e-2 * abs(v) + 1e-5, (what, i, m[i], ma[i])


@pytest.mark.parametrize("B,H", [(1, 128), (2, 256)])
def test_deconv2_normalising_deconv1_in_its_own_kernel(monkeypatch, B, H):
    """The same switch on its own (engine.D4_XF), where the general kernel takes deconv2 (too few tiles for the halo
    kernel: one 128 x 128 slice) and where the halo kernel does: same prediction as the plan with the separate pass."""
    sd = g_state()
    x, y = synth_batch(B, H, seed=83)
    outs = []
    for on in (False, True):
        monkeypatch.setattr(engine, "D4_XF", on)
        e = InferenceEngine(sd, B, H, H)
        assert (getattr(e.G.last.core, "xf_mixed", None) is not None) == on
        p, _ = e.run(x.to(DEV), y.to(DEV))
        outs.append(p.cpu().clone())
    act_close(outs[1], outs[0], "deconv1 normalised inside deconv2 vs separate pass", mx=2e-2, mean=3e-3)


def test_training_plan_with_the_pool_branch_folded(monkeypatch):
    """RIED_POOL_FOLD_TRAIN=1: avg_pool + 1x1 of the pool branches as one 3x3 convolution also in training plans (off
    by default: measured slower).  Same step as the default plan up to bf16 rounding (the pooled tensor is no longer
    rounded; the weights are divided by nine before rounding)."""
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=97)
    base = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
    monkeypatch.setenv("RIED_POOL_FOLD_TRAIN", "1")
    fold = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
    assert any(isinstance(l, engine.ConvBN) and l.core.pool_fold for m in fold.G.body.layers for l in _walk(m))
    assert not any(isinstance(l, engine.ConvBN) and l.core.pool_fold for m in base.G.body.layers for l in _walk(m))
    for ts in (base, fold):
        ts.step(x.to(DEV), y.to(DEV))
    torch.cuda.synchronize()
    a, b = base.losses(), fold.losses()
    for k in ("dloss1", "dloss2", "gloss", "vgg", "dssim", "gen"):
        assert abs(a[k] - b[k]) <= 1e-2 * abs(a[k]) + 1e-4, (k, a[k], b[k])
    act_close(fold.G.out.cpu(), base.G.out.cpu(), "prediction", mx=2e-2, mean=3e-3)
    # the folded weights' gradients are the nine taps' sum over nine: a wrong scale shows in the norm ratio, a missing
    # tap in the cosine (these tensors sit in front of a BatchNorm and are small, so rounding alone costs ~0.15 of it)
    for k in ("layer_block_1.0.branch_pool.conv.weight", "layer_block_3.0.branch_pool.conv.weight"):
        grad_close(fold.gs.g[k].cpu(), base.gs.g[k].cpu(), "G " + k, cos=0.7, ratio=0.35)
    gf = torch.cat([fold.gs.g[k].float().flatten() for k in sorted(fold.gs.g)]).cpu()
    gb = torch.cat([base.gs.g[k].float().flatten() for k in sorted(base.gs.g)]).cpu()
    grad_close(gf, gb, "all generator gradients", cos=0.97, ratio=0.1)


@pytest.mark.parametrize("use_graph", [False, True])
def test_fresh_d_step_vs_oracle(use_graph):
    """fresh_d=True: the generator's adversarial term from a new discriminator forward on the prediction after both
    discriminator updates (the `D_gen = Dnet(pred)` commented out at src/train.py:171), against the oracle's
    restatement of that variant (bf16 storage emulated): losses 2 %, d gloss / d pred cosine 0.99, the spectral-norm
    vectors advanced three times.  It must differ from the default step in `gen` (otherwise the flag did nothing)."""
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=1234)
    ost = O.TrainState(gsd, dsd, vgg, lr=1e-4)
    with emu():
        ref = O.train_step(ost, x, y, want_grads=True, fresh_d=True)
        plain = O.train_step(O.TrainState(gsd, dsd, vgg, lr=1e-4), x, y)
    assert abs(ref["gen"] - plain["gen"]) > 1e-2 * abs(plain["gen"])
    ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=use_graph, fresh_d=True)
    assert ts.fresh_d and not ts.lookahead
    ts.step(x.to(DEV), y.to(DEV))
    got = ts.losses()
    for k in ("dloss1", "dloss2", "gloss", "vgg", "dssim", "gen"):
        assert abs(got[k] - ref[k]) <= 2e-2 * abs(ref[k]) + 1e-4, (k, got[k], ref[k])
    act_close(ts.G.out.cpu(), ref["pred"], "pred", mx=2e-2, mean=5e-3)
    grad_close(ts.dpred.cpu(), ref["dpred"], "d gloss / d pred", cos=0.99, ratio=0.03)
    for k in ("out_conv.weight", "deconv2.deconv.weight", "deconv1.deconv.weight"):
        grad_close(ts.gs.g[k].cpu(), ref["g_grads"][k], "G " + k, cos=0.95, ratio=0.08)
    for k in ("discriminator_net.0.conv2d.weight_u", "discriminator_net.6.conv2d.weight_v"):
        act_close(ts.ds.b[k].cpu(), ost.d[k], k, mx=5e-3, mean=5e-3)
    assert ts.gs.adam_step.item() == 1 and ts.ds.adam_step.item() == 2
    ts.close()
