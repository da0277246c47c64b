"""Network-level parity on the GPU: the HIP plans (bf16 storage, fp32 accumulate) against the CPU oracle on the
same seeded weights and inputs, and against the golden vectors made from the reference.

Two oracles are used.  "emu" is the oracle with its bf16 storage emulation on (every tensor the HIP path stores
in bf16 is rounded at the same point, arithmetic fp32): it isolates kernel correctness from storage precision,
and the tolerance against it is tight.  "fp32" is the plain reference arithmetic: the distance to it is the
cost of bf16 storage itself, which this network (32 train-mode BatchNorm layers deep, seeded random weights)
amplifies to a few percent on the deepest feature maps while the HU-scale output stays within 0.5 %.

The seeded generator used here is also chaotic in the dynamical sense: with batch 2 its 4x4 bottleneck
normalises over 32 samples per channel, and a perturbation grows about 3x per block (a 3e-4 relative difference
after reduce_grid_1, caused by a handful of bf16 roundings falling on the other side, is 1.4e-2 by layer_block_3
and 3.5e-2 by deconv1).  The same growth applies backwards to gradients.  Kernel correctness is therefore pinned
at block level (tests/test_blocks_gpu.py: every block, forward and backward, cosine 0.99999 against the
emulation); the whole-network checks below bound the accumulated drift.

Tolerances
  generator activations vs emu   mean |err| <= 8e-2 * mean|ref| (first four stages: 1e-2)
  generator activations vs fp32  mean |err| <= 1.5e-1 * mean|ref|
  generator output (HU)          mean |err| <= 5e-3 * mean|ref|, max <= 2e-2 * max|ref|  (emu, fp32 and golden)
  discriminator activations      mean |err| <= 3e-2 * mean|ref| vs emu
  losses                         rel err <= 2e-2 vs emu, 5e-2 vs fp32
  gradients vs emu               D: cosine >= 0.98 per tensor; G: cosine >= 0.95 for the last layers, >= 0.6
                                 for every tensor (drift through 30 BatchNorm layers, see above)
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ried_oracle as O
from reid_gan_amd import engine, ops
from reid_gan_amd.synth import synth_state, synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep, InferenceEngine

DEV = "cuda"
G_SEED, D_SEED = 101, 202


def nchw(v):
    return v.torch().float().cpu().permute(0, 3, 1, 2).contiguous()


class emu:
    """with emu(): the oracle rounds to bf16 wherever the HIP path stores bf16."""

    def __enter__(self):
        O.set_bf16_emulation(True)

    def __exit__(self, *a):
        O.set_bf16_emulation(False)


def act_close(got, ref, what, mx=1e-1, mean=3e-2):
    got, ref = got.double(), ref.double()
    e = (got - ref).abs()
    assert e.max().item() <= mx * ref.abs().max().item(), f"{what}: max err {e.max().item():.4g} vs {ref.abs().max().item():.4g}"
    assert e.mean().item() <= mean * ref.abs().mean().item() + 1e-12, f"{what}: mean err {e.mean().item():.4g} vs {ref.abs().mean().item():.4g}"


def grad_close(got, ref, what, cos=0.99, ratio=0.05):
    got, ref = got.double().flatten(), ref.double().flatten()
    ng, nr = got.norm().item(), ref.norm().item()
    if nr < 1e-12:
        assert ng < 1e-6, what
        return
    c = torch.dot(got, ref).item() / (ng * nr + 1e-30)
    assert c >= cos, f"{what}: cosine {c:.5f}"
    assert abs(ng / nr - 1) <= ratio, f"{what}: norm ratio {ng / nr:.4f}"


def g_state():
    return synth_state([(n, s) for n, s, _ in O.rigan_spec()], G_SEED)


def d_state():
    return synth_state([(n, s) for n, s, _ in O.discriminator_spec()], D_SEED)


def test_generator_forward_backward_vs_oracle_and_golden(golden_dir):
    sd = g_state()
    x, _ = synth_batch(2, 64, seed=7)
    # oracle (bf16 storage emulated) with autograd gradients, and the plain fp32 oracle
    osd = {k: (v.clone().requires_grad_(True) if O.trainable(k) else v.clone()) for k, v in sd.items()}
    P = O.Params(osd)
    acts = {}
    with emu():
        ref = O.rigan_forward(P, x, True, acts)
        loss = (ref / 500.0).square().mean()
        names = [k for k in osd if O.trainable(k)]
        grads = dict(zip(names, torch.autograd.grad(loss, [osd[k] for k in names])))
    dout = (2.0 / 500.0 ** 2 / ref.numel()) * ref.detach()
    acts32 = {}
    with torch.no_grad():
        ref32 = O.rigan_forward(O.Params({k: v.clone() for k, v in sd.items()}), x, True, acts32)
    # HIP
    st = engine.State(sd, DEV)
    G = engine.Generator(st, 2, 64, 64)
    out = G.forward(x.to(DEV))
    torch.cuda.synchronize()
    gold = np.load(os.path.join(golden_dir, "rigan_train_64.npz"))
    act_close(out.cpu(), ref.detach(), "out", mx=2e-2, mean=5e-3)
    act_close(out.cpu(), torch.from_numpy(gold["out"]), "out vs golden", mx=2e-2, mean=5e-3)
    act_close(out.cpu(), ref32, "out vs fp32", mx=2e-2, mean=5e-3)
    for name, slot in G.tops.items():
        early = name in ("conv1", "conv2", "layer_block_1", "reduce_grid_1")
        act_close(nchw(slot.v), acts[name].detach(), name, mx=0.3, mean=1e-2 if early else 8e-2)
        act_close(nchw(slot.v), acts32[name], name + " vs fp32", mx=1.0, mean=1.5e-1)
    for k in ("conv1.bn.running_mean", "conv1.bn.running_var", "layer_block_3.0.branch1x1.bn.running_var"):
        act_close(st.b[k].cpu(), P.sd[k], k, mx=2e-2, mean=1e-2)
    assert st.b["conv1.bn.num_batches_tracked"].item() == 1
    st.zero_grad()
    G.backward(dout.to(DEV).contiguous())
    torch.cuda.synchronize()
    bad = []
    for k in names:
        try:
            grad_close(st.g[k].cpu(), grads[k], k, cos=0.6, ratio=0.35)
        except AssertionError as e:
            bad.append(str(e))
    assert len(bad) <= len(names) // 25, bad[:10]
    for k in ("out_conv.weight", "deconv2.deconv.weight", "deconv2.bn.weight", "deconv1.deconv.weight"):
        grad_close(st.g[k].cpu(), grads[k], k, cos=0.95, ratio=0.08)


@pytest.mark.parametrize("grads", [True, False])
def test_sibling_groups_on_the_large_map_path(golden_dir, monkeypatch, grads):
    """Folded statistics off: every layer takes the path of large maps (one statistics row per convolution tile), where
    a sibling group's members finalize their channel slice of the shared rows (ried_bn_finalize with ldstat) - in the
    training plan and in the forward-only plan.  Against the golden output and against the same plan without groups."""
    monkeypatch.setattr(engine, "BN_FOLD", False)
    sliced_calls = []
    real = ops.bn_finalize

    def counting(*a, **kw):
        if kw.get("ldstat", 0) > 0:
            sliced_calls.append(kw["ldstat"])
        return real(*a, **kw)
    monkeypatch.setattr(ops, "bn_finalize", counting)
    sd = g_state()
    x, _ = synth_batch(2, 64, seed=7)
    dout = (torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(5)) * 1e-3).to(DEV)

    def run():
        st = engine.State(sd, DEV)
        G = engine.Generator(st, 2, 64, 64, grads=grads)
        out = G.forward(x.to(DEV)).cpu().clone()
        if grads:
            st.zero_grad()
            G.backward(dout)
        torch.cuda.synchronize()
        return out, st
    out, st = run()
    assert len(sliced_calls) >= 20, "no sibling group took the tile-row path"
    gold = np.load(os.path.join(golden_dir, "rigan_train_64.npz"))
    act_close(out, torch.from_numpy(gold["out"]), "out vs golden", mx=2e-2, mean=5e-3)
    monkeypatch.setattr(engine, "SIBLING_GROUPS", False)
    n = len(sliced_calls)
    out2, st2 = run()
    assert len(sliced_calls) == n
    act_close(out, out2, "grouped vs ungrouped", mx=1e-2, mean=2e-3)
    for k in st.b:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(st.b[k], st2.b[k], rtol=2e-2, atol=2e-3), k
    if grads:
        grad_close(st.flat_grad.cpu(), st2.flat_grad.cpu(), "gradients grouped vs ungrouped", cos=0.999, ratio=0.02)


def test_generator_eval_mode_and_batch_one(golden_dir):
    sd = g_state()
    x, _ = synth_batch(2, 64, seed=7)
    G = engine.Generator(engine.State(sd, DEV), 2, 64, 64, grads=False, training=False)
    out = G.forward(x.to(DEV)).cpu()
    act_close(out, torch.from_numpy(np.load(os.path.join(golden_dir, "rigan_eval_64.npz"))["out"]), "eval",
              mx=1e-1, mean=2e-2)      # seeded running stats do not match the data: activations reach 1e4
    x1, _ = synth_batch(1, 128, seed=8)
    G1 = engine.Generator(engine.State(sd, DEV), 1, 128, 128, grads=False, training=True)
    out1 = G1.forward(x1.to(DEV)).cpu()
    act_close(out1, torch.from_numpy(np.load(os.path.join(golden_dir, "rigan_train_128_b1.npz"))["out"]), "b1",
              mx=2e-2, mean=5e-3)


def test_discriminator_forward_backward_vs_oracle_and_golden(golden_dir):
    sd = d_state()
    xd, yd = synth_batch(2, 64, seed=9)
    d_in = torch.cat([yd, xd], 0)
    osd = {k: (v.clone().requires_grad_(True) if O.trainable(k) else v.clone()) for k, v in sd.items()}
    P = O.Params(osd)
    xin = d_in.clone().requires_grad_(True)
    acts = {}
    with emu():
        ref = O.discriminator_forward(P, xin, True, acts)
        pos, neg = torch.chunk(ref, 2, 0)
        dl = O.d_loss(pos, neg)
        names = [k for k in osd if O.trainable(k)]
        gr = torch.autograd.grad(dl, [osd[k] for k in names] + [xin])
    grads = dict(zip(names, gr[:-1]))
    st = engine.State(sd, DEV)
    D = engine.Discriminator(st, 4, 64, 64)
    out = D.forward(d_in.to(DEV))
    got = D.output_nchw().cpu()
    gold = np.load(os.path.join(golden_dir, "disc_train_64.npz"))
    for i in range(7):
        act_close(nchw(D.acts[i].v), acts[f"d{i}"].detach(), f"d{i}")
    act_close(got, ref.detach(), "out")
    act_close(got, torch.from_numpy(gold["out"]), "out vs golden (fp32 reference)", mx=1.0, mean=5e-2)
    for k in gold.files:
        if k.startswith("uv:"):
            act_close(st.b[k[3:]].cpu(), torch.from_numpy(gold[k]), k, mx=1e-4, mean=1e-4)
    # hinge on the batch means (src/loss.py:141-142): -1/n on pos and +1/n on neg where the hinge is active
    n = pos.numel()
    cpos = -1.0 / n if (1 - pos.mean()).item() > 0 else 0.0
    cneg = 1.0 / n if (1 + neg.mean()).item() > 0 else 0.0
    assert cpos != 0.0 or cneg != 0.0
    D.ctx.zero_backward(); st.zero_grad()
    dimg = torch.zeros(4, 1, 64, 64, device=DEV)
    ops.fill(out.g.images(0, 2), cpos)
    ops.fill(out.g.images(2, 4), cneg)
    D.backward(0, 4, wgrad=True, dimg=dimg)
    torch.cuda.synchronize()
    for k in names:
        if k.endswith("key_conv.bias"):     # softmax over keys is invariant to the key bias: the gradient is 0
            qn = grads[k.replace("key_conv", "query_conv")].norm().item()
            assert st.g[k].norm().item() <= 1e-2 * qn and grads[k].norm().item() <= 1e-2 * qn, k
            continue
        grad_close(st.g[k].cpu(), grads[k], k, cos=0.985, ratio=0.06)
    grad_close(dimg.cpu(), gr[-1], "d image", cos=0.985, ratio=0.06)
    # second forward advances u, v again
    D.forward(d_in.to(DEV))
    act_close(D.output_nchw().cpu(), torch.from_numpy(np.load(os.path.join(golden_dir, "disc_train_64_second.npz"))["out"]),
              "second forward vs golden (fp32 reference)", mx=1.0, mean=5e-2)


@pytest.mark.parametrize("B,H", [(2, 32), (1, 256)])      # 256 x 256: the patch size of the perceptual-heavy config
def test_vgg_taps_and_perceptual_gradient(B, H):
    vgg = synth_vgg()
    pred, y = synth_batch(B, H, seed=41)
    pr = pred.clone().requires_grad_(True)
    with emu():
        pn, gn = O.normalize_(pr), O.normalize_(y)
        fp = O.vgg_taps(vgg, torch.cat([pn, pn, pn], 1))
        fg = O.vgg_taps(vgg, torch.cat([gn, gn, gn], 1))
        vl = sum(torch.nn.functional.mse_loss(a, b) for a, b in zip(fp, fg)) / 5
        (gref,) = torch.autograd.grad(0.3 * vl, pr)
    V = engine.Vgg(vgg, B, H, H)
    V.forward(torch.cat([pred, y], 0).to(DEV))
    for i, (t, t32) in enumerate(V.taps):
        act_close(t32[:B].cpu().permute(0, 3, 1, 2), fp[i].detach(), f"tap{i} pred")
        act_close(t32[B:].cpu().permute(0, 3, 1, 2), fg[i].detach(), f"tap{i} target")
    loss = torch.zeros(1, device=DEV)
    dimg = torch.zeros(B, 1, H, H, device=DEV)
    V.ctx.zero_backward()
    V.loss_and_seed(loss, 0.3)
    V.backward(dimg)
    assert abs(loss.item() - vl.item()) <= 2e-2 * vl.item(), (loss.item(), vl.item())
    grad_close(dimg.cpu(), gref, "d perceptual / d pred", cos=0.98, ratio=0.08)
    # the fused route of the training step: target half first, then the prediction half with the MSE in the taps
    imgs = torch.cat([pred, y], 0).to(DEV)
    V.forward(imgs, B, 2 * B)
    V.forward_pred_fused(imgs, 0.3)
    loss2 = torch.zeros(1, device=DEV)
    V.loss_from_rows(loss2)
    dimg2 = torch.zeros(B, 1, H, H, device=DEV)
    V.backward(dimg2)
    assert abs(loss2.item() - loss.item()) <= 2e-3 * loss.item(), (loss2.item(), loss.item())
    grad_close(dimg2.cpu(), dimg.cpu(), "fused vs separate perceptual gradient", cos=0.999, ratio=0.01)


@pytest.mark.parametrize("use_graph", [False, True])
def test_train_step_vs_oracle(use_graph):
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=1234)
    ost = O.TrainState(gsd, dsd, vgg, lr=1e-4)
    with emu():
        ref = O.train_step(ost, x, y, want_grads=True)
    ost32 = O.TrainState(gsd, dsd, vgg, lr=1e-4)
    ref32 = O.train_step(ost32, x, y)
    ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=use_graph)
    ts.step(x.to(DEV), y.to(DEV))
    got = ts.losses()
    for k in ("dloss1", "dloss2", "gloss", "vgg", "dssim", "gen"):
        assert abs(got[k] - ref[k]) <= 2e-2 * abs(ref[k]) + 1e-4, (k, got[k], ref[k])
        assert abs(got[k] - ref32[k]) <= 5e-2 * abs(ref32[k]) + 1e-4, (k, got[k], ref32[k], "fp32")
    act_close(ts.G.out.cpu(), ref["pred"], "pred", mx=2e-2, mean=5e-3)
    act_close(ts.G.out.cpu(), ref32["pred"], "pred vs fp32", mx=2e-2, mean=5e-3)
    # gradients of the last D iteration and of G (flat_grad still holds them)
    for k, g in ref["d_grads_1"].items():
        if not k.endswith("key_conv.bias"):          # exactly zero in exact arithmetic
            grad_close(ts.ds.g[k].cpu(), g, "D " + k, cos=0.98, ratio=0.08)
    grad_close(ts.dpred.cpu(), ref["dpred"], "d gloss / d pred", cos=0.99, ratio=0.03)
    bad = []
    for k, g in ref["g_grads"].items():
        try:
            grad_close(ts.gs.g[k].cpu(), g, "G " + k, cos=0.6, ratio=0.35)
        except AssertionError as e:
            bad.append(str(e))
    assert len(bad) <= len(ref["g_grads"]) // 25, bad[:10]
    for k in ("out_conv.weight", "deconv2.deconv.weight", "deconv1.deconv.weight"):
        grad_close(ts.gs.g[k].cpu(), ref["g_grads"][k], "G " + k, cos=0.95, ratio=0.08)
    # Adam moved every parameter by about lr in the direction the oracle moved it
    for st, ost_sd, sd0 in ((ts.ds, ost.d, dsd), (ts.gs, ost.g, gsd)):
        for k in ("discriminator_net.6.conv2d.weight_orig", "out_conv.weight", "deconv1.deconv.weight"):
            if k in st.p:
                d_got = st.p[k].cpu() - sd0[k]
                d_ref = ost_sd[k] - sd0[k]
                agree = (torch.sign(d_got) == torch.sign(d_ref)).float().mean().item()
                assert agree > 0.9, (k, agree)
    # a second and third step run on the updated state
    ts.step()
    ts.step()
    l3 = ts.losses()
    with emu():
        O.train_step(ost, x, y)
        r3 = O.train_step(ost, x, y)
    O.train_step(ost32, x, y)
    f3 = O.train_step(ost32, x, y)
    # The first Adam step moves every parameter by +-lr whatever the size of its gradient, so after three steps
    # the fp32 oracle and its bf16-storage emulation are themselves ~15 % apart in dloss2: the engine has to land
    # within 10 % of the emulation or inside that band (widened by half).
    for k in ("dloss1", "dloss2", "gloss"):
        assert np.isfinite(l3[k])
        tol = max(0.1 * abs(r3[k]), 1.5 * abs(r3[k] - f3[k])) + 1e-3
        assert abs(l3[k] - r3[k]) <= tol, (k, l3[k], r3[k], f3[k])
    assert ts.gs.adam_step.item() == 3 and ts.ds.adam_step.item() == 6


def test_train_step_at_128_patches():
    """The reference's default patch size (train.py:44): larger maps through the tile-based kernels."""
    B, H = 2, 128
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=4321)
    with emu():
        ref = O.train_step(O.TrainState(gsd, dsd, vgg, lr=1e-4), x, y)
    ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
    ts.step(x.to(DEV), y.to(DEV))
    got = ts.losses()
    for k in ("dloss1", "dloss2", "gloss", "vgg", "dssim", "gen"):
        assert abs(got[k] - ref[k]) <= 2e-2 * abs(ref[k]) + 1e-4, (k, got[k], ref[k])
    act_close(ts.G.out.cpu(), ref["pred"], "pred", mx=2e-2, mean=5e-3)


def test_train_step_at_the_benchmark_batch():
    """Batch 128 of 64 x 64 (BASELINE configs[1], what bench.py times): the plan picks other convolution tiles,
    BatchNorm statistics folding, split-K factors and grouped weight-gradient shapes than at batch 4.

    Losses, prediction, d gloss / d pred and the discriminator's gradients: against the oracle with bf16 storage
    emulated, at the tolerances of the small-batch test.  Generator gradients: bf16 storage makes them noisy on these
    seeded weights whatever computes them - the oracle's own emulation and its fp32 run agree only to cosine 0.63-1.0
    per tensor (mean 0.88) at this batch - so every tensor is held to a bound calibrated by that pair: the engine must
    be at least as close to the emulation as the fp32 run is (minus 0.1), or above 0.95 (at most three of the 222
    tensors may miss that, none may fall below 0.5), with its norm within 30 % (or twice the fp32 run's own
    deviation); a mis-wired layer (cosine ~ 0, or a norm off by a factor) fails either way, and so does an engine that
    is on average further from the emulation than the fp32 run is.
    Measured: mean cosine 0.933-0.935 against 0.880 for the fp32 run, closer than it on 219-221 of 222 tensors
    (the engine's sums are folded with atomics, so its result moves a little from run to run)."""
    B, H = 128, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=1234)
    with emu():
        ref = O.train_step(O.TrainState(gsd, dsd, vgg, lr=1e-4), x, y, want_grads=True)
    f32 = O.train_step(O.TrainState(gsd, dsd, vgg, lr=1e-4), x, y, want_grads=True)
    # the plan bench.py times on one GPU: captured graph, look-ahead (the target-only work in front of this first
    # step, and again - for the same batch - at its end)
    ts = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True, lookahead=True)
    assert ts.lookahead and ts.pre_dreal
    ts.step(x.to(DEV), y.to(DEV))
    got = ts.losses()
    for k in ("dloss1", "dloss2", "gloss", "vgg", "dssim", "gen"):
        assert abs(got[k] - ref[k]) <= 2e-2 * abs(ref[k]) + 1e-4, (k, got[k], ref[k])
    act_close(ts.G.out.cpu(), ref["pred"], "pred", mx=2e-2, mean=5e-3)
    grad_close(ts.dpred.cpu(), ref["dpred"], "d gloss / d pred", cos=0.99, ratio=0.03)
    for k, g in ref["d_grads_1"].items():
        if not k.endswith("key_conv.bias"):
            grad_close(ts.ds.g[k].cpu(), g, "D " + k, cos=0.98, ratio=0.08)

    def cos_ratio(a, b):
        a, b = a.double().flatten(), b.double().flatten()
        return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)).item(), (a.norm() / (b.norm() + 1e-300)).item()
    bad, soft, closer, rows = [], [], 0, []
    for k, g in ref["g_grads"].items():
        c_eng, r_eng = cos_ratio(ts.gs.g[k].cpu(), g)
        c_ref, r_ref = cos_ratio(f32["g_grads"][k], g)
        rows.append((c_eng, c_ref, r_eng, r_ref, k))
        closer += c_eng >= c_ref
        # norm: within 30 % (or twice the fp32 run's own deviation); a tensor on which the two oracle runs only agree
        # to cosine c is noise to that degree, and noise adds to the norm: (1 - c) more is allowed
        row = (k, round(c_eng, 4), round(c_ref, 4), round(r_eng, 3), round(r_ref, 3))
        if abs(r_eng - 1) > max(0.3, 2.0 * abs(r_ref - 1)) + (1 - min(c_ref, 1.0)):
            bad.append(row)
        # direction: a mis-wired layer gives cosine ~ 0 and fails outright, as does a tensor the two oracle runs agree
        # on; on the noise-dominated ones (the sums are folded with atomics, so the engine's own result moves from run
        # to run) a few may fall below their individual bound
        elif c_eng < 0.5 or (c_ref >= 0.95 and c_eng < 0.9):
            bad.append(row)
        elif c_eng < min(0.95, c_ref - 0.1):
            soft.append(row)
    rows.sort()
    print("worst generator gradients (cos engine-emu, cos fp32-emu, norm ratios):", rows[:8])
    print("mean cosine engine-emu %.4f, fp32-emu %.4f; engine closer on %d of %d tensors"
          % (sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows), closer, len(rows)))
    assert not bad, (len(bad), bad[:12])
    assert len(soft) <= 3, (len(soft), soft)
    assert sum(r[0] for r in rows) >= sum(r[1] for r in rows), "on average further from the emulation than fp32 is"
    for k in ("out_conv.weight", "deconv2.deconv.weight", "deconv1.deconv.weight"):
        grad_close(ts.gs.g[k].cpu(), ref["g_grads"][k], "G " + k, cos=0.95, ratio=0.08)


def test_inference_at_the_benchmark_size_vs_oracle():
    """Batch 32 of 512 x 512 (BASELINE configs[2], what bench.py times) against the oracle's fp32 forward of the
    whole batch and its restatement of compute_measure on every slice."""
    sd = g_state()
    B, H = 32, 512
    x, y = synth_batch(B, H, seed=78)
    eng = InferenceEngine(sd, B, H, H)
    pred, sums = eng.run(x.to(DEV), y.to(DEV))
    p, got = pred.cpu().clone(), eng.measures(sums)
    del eng
    for i in (0, 13, 31):      # the metric kernel on the device prediction: fp32 both sides
        xi, yi, pi = [O.trunc(t[i, 0]) for t in (x, y, p)]
        o, q = O.compute_measure(xi, yi, pi, 400.0)
        for a, b in zip(got[i][0] + got[i][1], o + q):
            assert abs(a - b) <= 2e-4 * abs(b) + 2e-5, (i, got[i], o, q)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    pref, res, _ = O.infer_slice(sd, x, y)
    act_close(p, pref, "pred", mx=2e-2, mean=5e-3)
    for i in range(B):
        for a, b in zip(got[i][1], res[i][1]):
            assert abs(a - b) <= 2e-2 * abs(b), (i, got[i][1], res[i][1])


def weights_agree(a, b, steps=2):
    """Two runs of the same `steps` steps.  The few float atomics left in the step (loss sums) differ in the last bit
    between runs, and Adam's first updates are lr * sign(gradient)-like: a weight whose gradient is almost zero can
    move the other way (up to 2 * lr per step apart: lr 1e-4 for G, 4e-4 for D).  So: nearly all weights agree
    tightly, and none is further apart than those sign flips allow."""
    for st_a, st_b, lr in ((a.gs, b.gs, 1e-4), (a.ds, b.ds, 4e-4)):
        d = (st_a.flat - st_b.flat).abs()
        assert (d > 1.5 * steps * lr).float().mean().item() <= 1e-4, (lr, d.max().item())
        assert d.max().item() <= 2 * steps * lr * 1.05, (lr, d.max().item())


def test_segmented_graphs_equal_single_graph(monkeypatch):
    """With several ranks the step is captured as graph segments split at the collectives (and the side streams
    must rejoin before each split).  Here the collectives are stubbed out (identical replicas: the mean over ranks
    is the identity), so the segmented capture must reproduce the single-graph step."""
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=99)
    monkeypatch.setattr(engine, "BN_FOLD", False)        # fixed-order reductions: the two runs must agree closely
    one = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    seg = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    seg.segmented = True
    for ts in (one, seg):
        ts.step(x.to(DEV), y.to(DEV))
        ts.step()
    torch.cuda.synchronize()
    assert len(seg.graphs) > 3 and len(one.graphs) == 1
    a, b = one.losses(), seg.losses()
    for k in a:
        assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-6, (k, a[k], b[k])
    weights_agree(one, seg)


def test_train_step_through_a_one_rank_rccl_group(monkeypatch):
    """The data-parallel form of the step on one GPU: a process group of one rank (backend nccl = RCCL), so the
    graph segments are replayed with real all-reduces issued between them on the same stream.  The mean over one
    rank is the identity: losses and weights must equal the single-graph step."""
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    B, H = 4, 64
    gsd, dsd, vgg = g_state(), d_state(), synth_vgg()
    x, y = synth_batch(B, H, seed=98)
    monkeypatch.setattr(engine, "BN_FOLD", False)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        one = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
        par = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True, process_group=dist.group.WORLD)
        for ts in (one, par):
            ts.step(x.to(DEV), y.to(DEV))
            ts.step()
        torch.cuda.synchronize()
        assert par.world == 1 and len(par.graphs) > 3 and len(one.graphs) == 1
        a, b = one.losses(), par.losses()
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]) + 1e-6, (k, a[k], b[k])
        weights_agree(one, par)
    finally:
        dist.destroy_process_group()


def test_slice_inference_is_the_batch_one_loop():
    """SliceInference: every slice normalised with its own statistics, as the reference's batch-1 evaluation loop
    does - against the batch-1 engine slice by slice, against the oracle's forward of each slice alone, and unlike
    the whole-batch engine."""
    from reid_gan_amd.train_step import SliceInference
    sd = g_state()
    B, H = 5, 64
    x, y = synth_batch(B, H, seed=55)
    x[1] = x[1] * 0.25 - 600.0                  # slices with different statistics: batch and slice normalisation differ
    x[3] = x[3] * 1.5 + 300.0
    sl = SliceInference(sd, B, H, H, lanes=3)
    pred, sums = sl.run(x.to(DEV), y.to(DEV))
    pred, got = pred.cpu().clone(), sl.measures(sums)
    one = InferenceEngine(sd, 1, H, H)
    for i in range(B):
        p1, s1 = one.run(x[i:i + 1].to(DEV), y[i:i + 1].to(DEV))
        act_close(pred[i:i + 1], p1.cpu(), f"slice {i} vs the batch-1 engine", mx=5e-3, mean=5e-4)
        m1 = one.measures(s1)[0]
        for a, b in zip(got[i][1], m1[1]):
            assert abs(a - b) <= 2e-3 * abs(b) + 1e-5, (i, got[i], m1)
        with torch.no_grad():
            ref = O.rigan_forward(O.Params({k: v.clone() for k, v in sd.items()}), x[i:i + 1], True)
        act_close(pred[i:i + 1], ref, f"slice {i} vs the oracle's batch-1 forward", mx=2e-2, mean=5e-3)
    whole, _ = InferenceEngine(sd, B, H, H).run(x.to(DEV), y.to(DEV))
    assert (whole.cpu() - pred).abs().mean() > 10 * (pred[0:1] - one.run(x[0:1].to(DEV), y[0:1].to(DEV))[0].cpu()).abs().mean()
    # replays reuse the plans
    pred2, _ = sl.run(x.to(DEV), y.to(DEV))
    act_close(pred2.cpu(), pred, "second run", mx=5e-3, mean=5e-4)


def test_per_slice_statistics_inside_the_batched_kernels():
    """InferenceEngine(per_slice_stats=True): one batched pass whose BatchNorm layers normalise every slice with its
    own statistics - the reference's batch-1 evaluation loop.  Against the batch-1 engine slice by slice (same
    kernels, same arithmetic up to the order of the statistics sums), against the oracle's forward of each slice
    alone, and unlike the whole-batch statistics."""
    sd = g_state()
    B, H = 3, 256
    x, y = synth_batch(B, H, seed=56)
    x[1] = x[1] * 0.25 - 600.0
    x[2] = x[2] * 1.5 + 300.0
    eng = InferenceEngine(sd, B, H, H, per_slice_stats=True)
    pred, sums = eng.run(x.to(DEV), y.to(DEV))
    pred, got = pred.cpu().clone(), eng.measures(sums)
    one = InferenceEngine(sd, 1, H, H)
    worst = 0.0
    for i in range(B):
        p1, s1 = one.run(x[i:i + 1].to(DEV), y[i:i + 1].to(DEV))
        act_close(pred[i:i + 1], p1.cpu(), f"slice {i} vs the batch-1 engine", mx=5e-3, mean=5e-4)
        worst = max(worst, (pred[i:i + 1] - p1.cpu()).abs().mean().item())
        m1 = one.measures(s1)[0]
        for a, b in zip(got[i][1], m1[1]):
            assert abs(a - b) <= 2e-3 * abs(b) + 1e-5, (i, got[i], m1)
        with torch.no_grad():
            ref = O.rigan_forward(O.Params({k: v.clone() for k, v in sd.items()}), x[i:i + 1], True)
        act_close(pred[i:i + 1], ref, f"slice {i} vs the oracle's batch-1 forward", mx=2e-2, mean=5e-3)
    whole, _ = InferenceEngine(sd, B, H, H).run(x.to(DEV), y.to(DEV))
    assert (whole.cpu() - pred).abs().mean().item() > 10 * worst + 1e-3
    with pytest.raises(ValueError):
        InferenceEngine(sd, 2, 64, 64, per_slice_stats=True)          # maps too small for whole pixel tiles


def test_inference_metrics_vs_oracle():
    sd = g_state()
    B, H = 2, 128
    x, y = synth_batch(B, H, seed=77)
    eng = InferenceEngine(sd, B, H, H)
    pred, sums = eng.run(x.to(DEV), y.to(DEV))
    got = eng.measures(sums)
    # metrics of the device prediction, computed by the oracle's restatement of measure.py
    p = pred.cpu()
    for i in range(B):
        xi, yi, pi = [O.trunc(t[i, 0]) for t in (x, y, p)]
        o, q = O.compute_measure(xi, yi, pi, 400.0)
        for a, b in zip(got[i][0] + got[i][1], o + q):
            assert abs(a - b) <= 2e-4 * abs(b) + 2e-5, (got[i], o, q)
    # and end to end against the oracle's own forward (bf16 generator vs fp32 generator)
    pref, res, _ = O.infer_slice(sd, x, y)
    act_close(p, pref, "pred", mx=2e-2, mean=5e-3)
    for i in range(B):
        for a, b in zip(got[i][1], res[i][1]):
            assert abs(a - b) <= 2e-2 * abs(b), (got[i][1], res[i][1])


def test_inference_at_the_benchmark_size_with_and_without_sibling_groups(monkeypatch):
    """Batch 32 of 512 x 512 (the inference workload of the benchmark): the sibling groups' shared convolutions write
    thousands of statistics rows there, which every member reduces in two steps from its channel slice.  The
    prediction and the per-slice metrics must equal the plan without groups up to bf16 rounding of the shared
    convolution's different tile order."""
    sd = g_state()
    B, H = 32, 512
    x, y = synth_batch(B, H, seed=78)
    xd, yd = x.to(DEV), y.to(DEV)
    a = InferenceEngine(sd, B, H, H)
    pa, sa = a.run(xd, yd)
    pa, ma = pa.cpu().clone(), a.measures(sa)
    del a
    monkeypatch.setattr(engine, "SIBLING_GROUPS", False)
    b = InferenceEngine(sd, B, H, H)
    pb, sb = b.run(xd, yd)
    pb, mb = pb.cpu(), b.measures(sb)
    assert torch.isfinite(pa).all()
    act_close(pa, pb, "grouped vs ungrouped prediction", mx=1e-2, mean=1e-3)
    for i in range(B):
        for u, v in zip(ma[i][1], mb[i][1]):
            assert abs(u - v) <= 5e-3 * abs(v) + 1e-5, (i, ma[i], mb[i])


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
    its own kernel (ried_conv2d_xf; off by default: measured slower).  Same prediction and metrics as the plan that
    writes every normalised tensor, up to bf16 rounding of tensors that are no longer stored."""
    sd = g_state()
    B, H = 4, 256
    x, y = synth_batch(B, H, seed=79)
    xd, yd = x.to(DEV), y.to(DEV)
    a = InferenceEngine(sd, B, H, H)
    pa, sa = a.run(xd, yd)
    pa, ma = pa.cpu().clone(), a.measures(sa)
    monkeypatch.setattr(engine, "XF_FUSE", True)
    b = InferenceEngine(sd, B, H, H)
    fused = [l for m in b.G.body.layers for l in _walk(m) if isinstance(l, engine.ConvBN) and l.core.xf_src is not None]
    assert len(fused) >= 10, len(fused)
    pb, sb = b.run(xd, yd)
    pb, mb = pb.cpu(), b.measures(sb)
    act_close(pb, pa, "fused vs separate normalisation", mx=2e-2, mean=3e-3)
    for i in range(B):
        for u, v in zip(mb[i][1], ma[i][1]):
            assert abs(u - v) <= 1e-2 * abs(v) + 1e-5, (i, mb[i], ma[i])


def _walk(layer):
    """Every leaf layer under a Seq / Par."""
    if isinstance(layer, engine.Seq):
        for l in layer.layers:
            yield from _walk(l)
    elif isinstance(layer, engine.Par):
        for b in layer.branches:
            yield from _walk(b)
    else:
        yield layer


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
