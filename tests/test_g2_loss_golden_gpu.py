"""The HIP loss path and the fused training step against fixtures made from the reference's own src/loss.py and from
its modules composed as src/train.py:153-177 composes them (tests/golden/make_golden_loss.py; torchvision replaced
by a three-name stand-in there, VGG16 weights seeded synthetic).

Tolerances
  compute_SSIM / SSIM (fp32 kernels)  against the reference code evaluated in fp64: 2e-4 + the reference's own
                                      fp32-vs-fp64 distance (1e-7 where E[x^2] - mu^2 is well conditioned; 0.08 at
                                      noise sd 0.5 and 6e-4 at sd 5 on a nearly constant region at -1000 HU);
                                      against the reference's fp32 value: 2e-4 + 1.5 x that distance
  DSSIM gradient                      within 2 % (norm of the difference) of the fp64 gradient in every regime
  DLoss / GLoss                       1e-5 relative (fp32 reductions)
  CompoundLoss, MPL (bf16 VGG)        loss 2 % (the perceptual part and MPL 5 %), gradient cosine >= 0.97
  training step (bf16 G, D, VGG)      losses 5 % against the fp32 reference, pred mean err 0.5 %, dpred cosine 0.98
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from reid_gan_amd import loss
from reid_gan_amd.synth import synth_state, synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep
from golden_loss_util import load_loss_json, load_npz, ssim_inputs, ssim_tolerance, cosine

RES = load_loss_json()
DEV = "cuda"


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def test_normalize_and_window():
    n = RES["normalize"]
    got = loss.normalize_(torch.tensor(n["in"], device=DEV)).cpu()
    assert np.allclose(got.numpy(), n["out"], rtol=0, atol=1e-7)
    assert np.allclose(loss.gaussian(11, 1.5).numpy(), RES["window_1d"], rtol=1e-7)
    assert np.allclose(loss.create_window(11, 1)[0, 0, 5].cpu().numpy(), RES["window_2d_row5"], rtol=1e-6)


@pytest.mark.parametrize("name", list(RES["ssim"]))
def test_compute_ssim_and_dssim_per_regime(name):
    """Every fixture holds the reference's fp32 result and the same reference code evaluated in fp64.  Where
    E[x^2] - mu^2 is well conditioned the two agree to 1e-7 and the kernel must match them to 2e-4.  Where it cancels
    (nearly constant region at -1000 HU) the reference's fp32 result is itself round-off (value 0.08 off at noise sd
    0.5, gradient norm 24 against a true 0.004): there the kernel is held to the fp64 value, which it must reach at
    least as closely as the reference's fp32 does, and to the fp32 value within that same distance."""
    c = RES["ssim"][name]
    t, p = ssim_inputs(c)
    g = load_npz("loss_grads.npz")
    ref_err = abs(c["ssim_mean"] - c["ssim_mean_fp64"])
    got = loss.compute_SSIM(t.to(DEV), p.to(DEV), 11, 1, True).item()
    assert abs(got - c["ssim_mean_fp64"]) <= 2e-4 + ref_err, (got, c["ssim_mean_fp64"], ref_err)
    assert abs(got - c["ssim_mean"]) <= ssim_tolerance(c), (got, c["ssim_mean"], ssim_tolerance(c))
    per = loss.compute_SSIM(t.to(DEV), p.to(DEV), 11, 1, False).cpu().numpy()
    assert per.shape == (c["batch"],)
    assert np.all(np.abs(per - np.array(c["ssim_per_image"])) <= 2 * ssim_tolerance(c)), (per, c["ssim_per_image"])
    pr = p.to(DEV).requires_grad_(True)
    d = loss.SSIM()(t.to(DEV), pr)
    d.backward()
    assert abs(d.item() - (1 - c["ssim_mean_fp64"]) / 2) <= 1e-4 + ref_err / 2
    g64 = torch.from_numpy(g["dssim_grad64:" + name]).double()
    g32 = torch.from_numpy(g["dssim_grad:" + name]).double()
    err = (pr.grad.cpu().double() - g64).norm().item() / g64.norm().item()
    err_ref = (g32 - g64).norm().item() / g64.norm().item()
    # gradient of the mean SSIM against the fp64 evaluation: 2 % of its norm, in every regime
    assert err <= 2e-2, (name, err, err_ref)
    if err_ref < 1e-3:       # and against the reference's fp32 gradient wherever that is itself meaningful
        assert cosine(pr.grad, g32) >= 0.999 and _rel(pr.grad.double().norm().item(), c["dssim_grad_norm"]) <= 0.02


def test_hinge_losses():
    g = torch.Generator().manual_seed(77)
    for h, (sp, sn) in zip(RES["hinge"], [(0.3, -0.2), (1.7, -0.2), (0.3, -1.9), (2.5, -3.0), (-0.5, 0.8)]):
        pos = (torch.randn(4, 8192, generator=g) + sp).to(DEV).requires_grad_(True)
        neg = (torch.randn(4, 8192, generator=g) + sn).to(DEV).requires_grad_(True)
        dl = loss.DLoss()(pos, neg)
        dl.backward()
        assert abs(dl.item() - h["dloss"]) <= 1e-5 * max(1.0, abs(h["dloss"])), (dl.item(), h)
        assert abs(loss.GLoss()(neg.detach()).item() - h["gloss"]) <= 1e-5 * abs(h["gloss"])
        assert abs(pos.grad[0, 0].item() - h["dpos"]) <= 1e-9 and abs(neg.grad[0, 0].item() - h["dneg"]) <= 1e-9


def test_compound_loss_against_reference():
    c = RES["compound"]
    g = load_npz("loss_grads.npz")
    vgg = synth_vgg()
    pred, y = synth_batch(c["batch"], c["size"], seed=c["seed"])
    d_out = torch.randn(c["batch"], 64, generator=torch.Generator().manual_seed(c["d_seed"]))
    crit = loss.CompoundLoss(vgg_state=vgg)
    p, d = pred.to(DEV).requires_grad_(True), d_out.to(DEV).requires_grad_(True)
    got = crit(p, y.to(DEV), d)
    got.backward()
    terms = crit.last_terms.cpu()
    assert _rel(got.item(), c["loss"]) <= 2e-2, (got.item(), c["loss"])
    assert _rel(terms[5].item(), c["vgg"]) <= 5e-2, (terms[5].item(), c["vgg"])
    assert _rel(terms[7].item(), c["dssim"]) <= 2e-3, (terms[7].item(), c["dssim"])
    assert _rel(terms[8].item(), c["gen"]) <= 1e-5
    assert cosine(p.grad, g["compound:dpred"]) >= 0.97, cosine(p.grad, g["compound:dpred"])
    assert _rel(p.grad.double().norm().item(), c["dpred_norm"]) <= 0.08
    assert abs(d.grad[0, 0].item() - c["dd_out"]) <= 1e-8
    taps = loss.Vgg16FeatureExtractor(vgg_state=vgg)(
        torch.cat([loss.normalize_(pred)] * 3, 1).repeat(2, 1, 1, 1).to(DEV))
    assert [list(t.shape) for t in taps] == [[2 * s[0]] + s[1:] for s in c["tap_shapes"]]
    for i, t in enumerate(taps):
        half = t[:c["batch"]].detach().reshape(-1).cpu()
        stride = max(1, half.numel() // 256)
        got_s, ref_s = half[::stride][:256].numpy(), g[f"compound:tap{i}"]
        assert np.abs(got_s - ref_s).mean() <= 3e-2 * np.abs(ref_s).mean() + 1e-6, i


def test_mpl_against_reference():
    c = RES["mpl"]
    g = load_npz("loss_grads.npz")
    vgg = synth_vgg()
    pred, y = synth_batch(c["batch"], c["size"], seed=c["seed"])
    crit = loss.MPL(vgg_state=vgg)
    p = pred.to(DEV).requires_grad_(True)
    got = crit(y.to(DEV), p)
    got.backward()
    # MPL feeds RAW HU (no normalize_, loss.py:36) through the bf16 VGG: the features are ~1e3 with rounding steps of
    # a few units, and the loss is a norm of small feature DIFFERENCES, which that rounding noise inflates (measured
    # +3.3 % against the fp32 reference; 1 % against the oracle's bf16 emulation in test_modules_gpu.py)
    assert _rel(got.item(), c["loss"]) <= 5e-2, (got.item(), c["loss"])
    assert cosine(p.grad, g["mpl:dpred"]) >= 0.97
    assert _rel(p.grad.double().norm().item(), c["dpred_norm"]) <= 0.06


@pytest.mark.parametrize("use_graph", [False, True])
def test_train_step_against_reference_composition(use_graph):
    """BASELINE configs[0] (batch 4 of 1x64x64) through the fused HIP step: the losses of three consecutive steps,
    the prediction and the gradients of the first step against the reference's modules + torch.optim.Adam (fp32)."""
    from oracle import ried_oracle as O
    s = load_npz("step_64.npz")
    gsd = synth_state([(n, sh) for n, sh, _ in O.rigan_spec()], 101)
    dsd = synth_state([(n, sh) for n, sh, _ in O.discriminator_spec()], 202)
    x, y = synth_batch(4, 64, seed=1234)
    ts = TrainStep(gsd, dsd, synth_vgg(), 4, 64, 64, lr=1e-4, use_graph=use_graph)
    ts.step(x.to(DEV), y.to(DEV))
    got = ts.losses()
    ref = s["losses"]
    assert _rel(got["dloss1"], ref[0, 0]) <= 5e-2, (got, ref[0])
    assert _rel(got["dloss2"], ref[0, 1]) <= 5e-2, (got, ref[0])
    assert _rel(got["gloss"], ref[0, 2]) <= 5e-2, (got, ref[0])
    assert _rel(got["vgg"], s["vgg"][0]) <= 5e-2 and _rel(got["dssim"], s["dssim"][0]) <= 5e-2
    assert _rel(got["gen"], s["gen"][0]) <= 5e-2
    pred = ts.G.out.cpu()
    e = (pred.double() - torch.from_numpy(s["pred"]).double()).abs()
    assert e.mean().item() <= 5e-3 * np.abs(s["pred"]).mean() and e.max().item() <= 2e-2 * np.abs(s["pred"]).max()
    assert cosine(ts.dpred, s["dpred"]) >= 0.98, cosine(ts.dpred, s["dpred"])
    # gradients: the last D iteration's and the generator's are still in the flat buffers
    for k in ts.ds.g:
        if k.endswith("key_conv.bias"):
            continue
        refn = s["dgn1:" + k][0]
        assert _rel(ts.ds.g[k].double().norm().item(), refn) <= 0.12, (k, ts.ds.g[k].norm().item(), refn)
    near = 0
    for k in ts.gs.g:
        refn = s["ggn:" + k][0]
        near += _rel(ts.gs.g[k].double().norm().item(), refn) <= 0.35
    assert near >= 0.9 * len(ts.gs.g), (near, len(ts.gs.g))       # batch 4: BatchNorm over 64 samples is chaotic
    for k in ("out_conv.weight", "deconv2.deconv.weight", "deconv1.deconv.weight"):
        assert _rel(ts.gs.g[k].double().norm().item(), s["ggn:" + k][0]) <= 0.1, k
    ts.step()
    ts.step()
    l3 = ts.losses()
    # three sign-like Adam steps (beta1 = 0.01) apart, bf16 and fp32 trajectories have drifted (the oracle's own bf16
    # emulation is 15 % from its fp32 run in dloss2 here): same regime, not the same digits
    for i, k in enumerate(("dloss1", "dloss2", "gloss")):
        assert np.isfinite(l3[k]) and _rel(l3[k], ref[2, i]) <= 0.3, (k, l3[k], ref[2, i])
