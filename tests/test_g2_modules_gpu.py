"""The drop-in module / loss / metric API on the GPU, used the way the reference's train.py and test.py use it."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ried_oracle as O
from reid_gan_amd import rinet, loss, measure
from reid_gan_amd.synth import synth_state, synth_batch, synth_vgg

DEV = "cuda"


def _load(m, seed):
    m.load_state_dict(synth_state(rinet.state_spec(m), seed))
    return m


def _rel(a, b):
    return (a.double() - b.double()).abs().mean().item() / (b.double().abs().mean().item() + 1e-30)


def test_rigan_module_against_golden(golden_dir):
    G = _load(rinet.RIGAN(), 101).cuda()
    x, _ = synth_batch(2, 64, seed=7)
    out = G(x.cuda())
    gold = np.load(os.path.join(golden_dir, "rigan_train_64.npz"))
    assert _rel(out.detach().cpu(), torch.from_numpy(gold["out"])) < 5e-3
    loss_v = (out / 500.0).square().mean()
    loss_v.backward()
    assert abs(loss_v.item() - gold["loss"][0]) < 1e-2 * gold["loss"][0]
    gw = G.out_conv.weight.grad
    ref = gold["gs:out_conv.weight"]
    assert _rel(gw.flatten()[::max(1, gw.numel() // 16)][:16].cpu(), torch.from_numpy(ref)) < 2e-2
    # running statistics were updated in the module's own buffers
    assert G.conv1.bn.num_batches_tracked.item() == 1
    assert _rel(G.conv1.bn.running_mean.cpu(), torch.from_numpy(gold["rs:conv1.bn.running_mean"])) < 1e-2
    # eval mode uses them
    G.eval()
    with torch.no_grad():
        o2 = G(x.cuda())
    assert o2.shape == out.shape and torch.isfinite(o2).all()


def test_discriminator_module_against_golden(golden_dir):
    D = _load(rinet.ImageDiscriminator(), 202).cuda()
    xd, yd = synth_batch(2, 64, seed=9)
    d_in = torch.cat([yd, xd], 0).cuda().requires_grad_(True)
    out = D(d_in)
    gold = np.load(os.path.join(golden_dir, "disc_train_64.npz"))
    assert out.shape == (4, 8192)
    assert _rel(out.detach().cpu(), torch.from_numpy(gold["out"])) < 2e-2
    pos, neg = torch.chunk(out, 2, 0)
    dl = loss.DLoss()(pos, neg)
    assert abs(dl.item() - gold["dloss"][0]) < 2e-2 * abs(gold["dloss"][0])
    dl.backward()
    assert _rel(d_in.grad.cpu(), torch.from_numpy(gold["dx"])) < 0.15
    g = D.discriminator_net[6].conv2d.weight_orig.grad
    assert abs(g.double().norm().item() / gold["gn:discriminator_net.6.conv2d.weight_orig"][0] - 1) < 0.05
    assert _rel(D.discriminator_net[0].conv2d.weight_u.cpu(),
                torch.from_numpy(gold["uv:discriminator_net.0.conv2d.weight_u"])) < 1e-4


@pytest.mark.parametrize("name,ctor,cin,hw,seed", [
    ("basic_conv_3x3s2", lambda: rinet.BasicConv2d(32, 64, 3, 2, padding=1), 32, 8, 300),
    ("basic_deconv_4x4s2", lambda: rinet.BasicDeconv2d(96, 32, 4, 2, padding=1), 96, 8, 302),
    ("inception_b_dec", lambda: rinet.InceptionB(256, 1), 256, 4, 307),
    ("inception_d_enc", lambda: rinet.InceptionD(256, None), 256, 8, 310),
    ("snconv_64_128_s2", lambda: rinet.SNConv2d(64, 128, 3, 2, padding=1), 64, 8, 313),
    ("self_attn_256", lambda: rinet.Self_Attn(256), 256, 8, 314),
    ("self_attn_512", lambda: rinet.Self_Attn(512), 512, 4, 315),
])
def test_block_modules_against_golden(golden_dir, name, ctor, cin, hw, seed):
    m = _load(ctor(), seed).cuda()
    bi = seed - 300
    r = np.random.default_rng(500 + bi)
    xin = torch.from_numpy(r.standard_normal((2, cin, hw, hw)).astype(np.float32))
    if "self_attn" not in name and "snconv" not in name:
        xin = xin.abs()
    x = xin.cuda().requires_grad_(True)
    y = m(x)
    gold = np.load(os.path.join(golden_dir, f"block_{name}.npz"))
    assert _rel(y.detach().cpu(), torch.from_numpy(gold["y"])) < 3e-2
    gw = torch.from_numpy(r.standard_normal(tuple(y.shape)).astype(np.float32)).cuda()
    (y * gw).sum().backward()
    a, b = x.grad.cpu().double().flatten(), torch.from_numpy(gold["dx"]).double().flatten()
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.97
    for k, p in m.named_parameters():
        ref = gold["gn:" + k][0]
        if name.startswith("self_attn") and k == "key_conv.bias":
            continue
        assert abs(p.grad.double().norm().item() / ref - 1) < 0.08, k


def test_identity_downsample_block_module(golden_dir):
    import torch.nn as nn
    ds = nn.Sequential(nn.Conv2d(64, 128, 1, 1, bias=False), nn.BatchNorm2d(128), nn.ReLU())
    m = _load(rinet.InceptionA(64, ds), 304).cuda()
    r = np.random.default_rng(504)
    x = torch.from_numpy(r.standard_normal((2, 64, 8, 8)).astype(np.float32)).abs().cuda()
    y = m(x)
    gold = np.load(os.path.join(golden_dir, "block_inception_a_64.npz"))
    assert _rel(y.detach().cpu(), torch.from_numpy(gold["y"])) < 3e-2


def test_losses_and_metrics_api(golden_dir):
    res = json.load(open(os.path.join(golden_dir, "measure.json")))
    for case in "abc":
        c = res[case]
        xm, ym = synth_batch(1, c["size"], seed=c["seed"])
        r = np.random.default_rng(c["seed"] + 1000)
        pm = ym + torch.from_numpy((c["noise"] * r.standard_normal(tuple(ym.shape))).astype(np.float32))
        xm, ym, pm = [t.reshape(c["size"], c["size"]).clamp(-160.0, 240.0) for t in (xm, ym, pm)]
        o, p = measure.compute_measure(xm, ym, pm, 400.0)          # CPU tensors in, as test.py passes them
        for a, b in zip(o + p, c["orig"] + c["pred"]):
            assert abs(a - b) <= 2e-4 * abs(b) + 2e-5, (case, o, p)
        assert abs(measure.compute_MSE(pm.numpy(), ym.numpy()) - c["mse"]) < 1e-4 * c["mse"]
        assert abs(measure.compute_CNR(pm, ym) - c["cnr_numpy"]) < 2e-3 * abs(c["cnr_numpy"]) + 1e-6
        assert abs(loss.compute_SSIM(xm.cuda(), ym.cuda(), 11, 1).item() - c["ssim_dr1_raw"]) < 3e-4
    xm, ym = synth_batch(4, 64, seed=31)
    assert abs(loss.compute_SSIM(ym.cuda(), xm.cuda(), 11, 1).item() - res["raw_b4"]["ssim_dr1"]) < 2e-4
    # SSIM module gradient wrt pred
    pred = xm.cuda().requires_grad_(True)
    l = loss.SSIM()(ym.cuda(), pred)
    l.backward()
    pr = xm.clone().requires_grad_(True)
    lr = O.dssim_loss(ym, pr)
    lr.backward()
    assert abs(l.item() - lr.item()) < 2e-4
    a, b = pred.grad.cpu().double().flatten(), pr.grad.double().flatten()
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.999


def test_compound_loss_against_oracle():
    vgg = synth_vgg()
    B, H = 2, 32
    pred, y = synth_batch(B, H, seed=41)
    d_out = torch.randn(B, 64, generator=torch.Generator().manual_seed(3))
    pr, dr = pred.clone().requires_grad_(True), d_out.clone().requires_grad_(True)
    O.set_bf16_emulation(True)
    try:
        ref, vl, sl, gl = O.compound_loss(vgg, pr, y, dr)
        ref.backward()
    finally:
        O.set_bf16_emulation(False)
    crit = loss.CompoundLoss(vgg_state=vgg)
    p, d = pred.cuda().requires_grad_(True), d_out.cuda().requires_grad_(True)
    got = crit(p, y.cuda(), d)
    got.backward()
    assert abs(got.item() - ref.item()) < 1e-2 * abs(ref.item())
    a, b = p.grad.cpu().double().flatten(), pr.grad.double().flatten()
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.99
    assert torch.allclose(d.grad.cpu(), dr.grad, rtol=1e-4, atol=1e-9)
    taps = loss.Vgg16FeatureExtractor(vgg_state=vgg)(torch.cat([O.normalize_(pred)] * 3, 1).repeat(2, 1, 1, 1).cuda())
    assert [t.shape[1] for t in taps] == [64, 128, 256, 512, 512]


def test_mpl_against_oracle():
    """MPL (src/loss.py:89-110) and get_feature_layer_vgg16 against the oracle's restatement (bf16 storage emulated);
    parity with the reference itself is unpinned: loss.py needs torchvision, which is not installed."""
    vgg = synth_vgg()
    B, H = 2, 32
    pred, y = synth_batch(B, H, seed=43)
    pr = pred.clone().requires_grad_(True)
    O.set_bf16_emulation(True)
    try:
        ref = O.mpl_loss(vgg, y, pr)
        ref.backward()
        tap8 = O.vgg_taps(vgg, torch.cat([torch.cat([pred, y], 0)] * 3, 1))[1]
    finally:
        O.set_bf16_emulation(False)
    crit = loss.MPL(vgg_state=vgg)
    p = pred.cuda().requires_grad_(True)
    got = crit(y.cuda(), p)
    got.backward()
    assert abs(got.item() - ref.item()) < 1e-2 * abs(ref.item()), (got.item(), ref.item())
    a, b = p.grad.cpu().double().flatten(), pr.grad.double().flatten()
    # the gradient is a field of unit directions d/|d|: rounding noise in small feature differences turns them
    # more than it changes the loss (measured cosine 0.986 at this size)
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.97
    assert abs(a.norm().item() / b.norm().item() - 1.0) < 0.05
    f = loss.get_feature_layer_vgg16(torch.cat([pred, y], 0).cuda(), 8, crit).cpu()
    assert f.shape == tap8.shape == (2 * B, 128, H // 2, H // 2)
    assert (f - tap8).abs().mean().item() <= 1e-2 * tap8.abs().mean().item()
    with pytest.raises(NotImplementedError):
        loss.get_feature_layer_vgg16(pred.cuda(), 5, crit)


def test_reference_training_loop_shape():
    """The body of the reference's train.py:153-177, verbatim in structure, on the drop-in modules with
    torch.optim.Adam: it runs (the reference's own modules raise here on torch >= 1.5) and the first
    iteration's losses match the oracle's statement of the step."""
    gsd = synth_state(rinet.state_spec(rinet.RIGAN()), 101)
    dsd = synth_state(rinet.state_spec(rinet.ImageDiscriminator()), 202)
    vgg = synth_vgg()
    Gnet, Dnet = rinet.RIGAN().cuda(), rinet.ImageDiscriminator().cuda()
    Gnet.load_state_dict(gsd); Dnet.load_state_dict(dsd)
    optG = torch.optim.Adam(Gnet.parameters(), lr=1e-4, betas=(0.01, 0.99))
    optD = torch.optim.Adam(Dnet.parameters(), lr=4e-4, betas=(0.01, 0.99))
    Dloss, criterion = loss.DLoss(), loss.CompoundLoss(vgg_state=vgg)
    x, y = synth_batch(4, 64, seed=1234)
    ost = O.TrainState(gsd, dsd, vgg)
    O.set_bf16_emulation(True)
    try:
        ref = O.train_step(ost, x, y)
    finally:
        O.set_bf16_emulation(False)
    x, y = x.cuda(), y.cuda()
    Dnet.train(); Gnet.train()
    for it in range(2):
        pred = Gnet(x)
        dl = []
        for _ in range(2):
            optD.zero_grad(); Dnet.zero_grad()
            pred_pos_neg = Dnet(torch.cat([y, pred], dim=0))
            pred_pos, pred_neg = torch.chunk(pred_pos_neg, 2, dim=0)
            dloss = Dloss(pred_pos, pred_neg).clone()
            dloss.backward(retain_graph=True)
            optD.step()
            dl.append(dloss.item())
        optG.zero_grad(); Gnet.zero_grad()
        gloss = criterion(pred, y, pred_neg)
        gloss.backward(retain_graph=True)
        optG.step()
        if it == 0:
            assert abs(dl[0] - ref["dloss1"]) < 2e-2 * abs(ref["dloss1"]), (dl, ref["dloss1"])
            assert abs(dl[1] - ref["dloss2"]) < 3e-2 * abs(ref["dloss2"]), (dl, ref["dloss2"])
            assert abs(gloss.item() - ref["gloss"]) < 2e-2 * abs(ref["gloss"]), (gloss.item(), ref["gloss"])
    assert np.isfinite(gloss.item())
    sd = Gnet.state_dict()
    assert set(sd) == set(gsd) and not torch.equal(sd["out_conv.weight"].cpu(), gsd["out_conv.weight"])


def test_backward_of_a_superseded_forward_is_refused():
    """A module's plan holds the activations of its last forward per input shape: the backward of an earlier
    forward of that shape raises instead of using the later call's activations."""
    torch.manual_seed(0)
    m = _load(rinet.BasicConv2d(16, 16, kernel_size=3, padding=1), 5).to(DEV)
    x1 = torch.randn(2, 16, 8, 8, device=DEV, requires_grad=True)
    x2 = torch.randn(2, 16, 8, 8, device=DEV, requires_grad=True)
    y1 = m(x1)
    y2 = m(x2)
    with pytest.raises(RuntimeError, match="forward call"):
        y1.sum().backward()
    y2.sum().backward()                              # the latest forward is fine
    assert x2.grad is not None and torch.isfinite(x2.grad).all()
    d = _load(rinet.ImageDiscriminator(), 202).to(DEV)
    a = torch.randn(2, 1, 64, 64, device=DEV)
    o1 = d(a)
    o2 = d(a + 1)
    with pytest.raises(RuntimeError, match="forward call"):
        o1.mean().backward()
    o2.mean().backward()


def test_psnr_of_identical_images_is_inf():
    """measure.py:29-32: log10(range^2 / 0) is inf in the reference, not an exception."""
    a = torch.rand(1, 1, 64, 64, device=DEV)
    assert measure.compute_PSNR(a, a.clone(), 1.0) == float("inf")
