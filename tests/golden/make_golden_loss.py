"""Golden vectors for src/loss.py and for the training step of src/train.py:153-177, made by importing the reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_loss.py

src/loss.py imports torchvision at module level (loss.py:8-10) and torchvision is not installed here.  The module
is imported with three names pre-seeded in sys.modules (test-only code below, never on the product path):

  torchvision.models.vgg16(pretrained, progress)          -> an object with .features = the VGG16 configuration-D
                                                             Sequential (13 Conv3x3 pad 1 + ReLU(inplace), 5 MaxPool
                                                             2x2; 31 modules) and deletable .avgpool / .classifier
  torchvision.models._utils.IntermediateLayerGetter       -> run named children in order, collect the named outputs,
                                                             drop the children after the last requested one
  torchvision.transforms.Normalize                        -> constructed at loss.py:79, never called

normalize_, gaussian, create_window, compute_SSIM, SSIM, DLoss and GLoss pass no arithmetic through those names, so
their fixtures are outputs of the reference's own code.  CompoundLoss, MPL and Vgg16FeatureExtractor run the
reference's composition (tap selection, normalisation, weighting) over the stand-in's VGG16 layer stack with seeded
synthetic weights: what stays unpinned is only "this layer stack == torchvision's" and the ImageNet weights.

The step fixture composes the reference's RIGAN / ImageDiscriminator / DLoss / CompoundLoss objects and
torch.optim.Adam exactly as train.py:153-177 does, with the one change that makes it executable on torch >= 1.5:
the generator gradient is taken with autograd.grad BEFORE the last optimizer_discriminator.step() (the verbatim
script raises at gloss.backward because that step modified D's weights in place).

Only numbers are written (tests/golden/loss.json, loss_grads.npz, step_64.npz); no reference source is copied.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch
import torch.nn as nn

from reid_gan_amd.synth import synth_state, synth_batch, synth_vgg

torch.set_num_threads(8)

# ------------------------------------------------------------------ the three-name torchvision stand-in
_CFG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


class _Vgg16(nn.Module):
    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _CFG_D:
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.Identity()
        self.classifier = nn.Identity()


class _IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        want = dict(return_layers)
        left = set(want)
        kept = {}
        for name, child in model.named_children():
            kept[name] = child
            left.discard(name)
            if not left:
                break
        super().__init__(kept)
        self.return_layers = want

    def forward(self, x):
        out = {}
        for name, child in self.items():
            x = child(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def _install_stand_in():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    utils = types.ModuleType("torchvision.models._utils")
    transforms = types.ModuleType("torchvision.transforms")
    models.vgg16 = lambda pretrained=False, progress=True, **kw: _Vgg16()
    utils.IntermediateLayerGetter = _IntermediateLayerGetter
    transforms.Normalize = lambda mean, std: None
    models._utils = utils
    tv.models, tv.transforms = models, transforms
    sys.modules.update({"torchvision": tv, "torchvision.models": models, "torchvision.models._utils": utils,
                        "torchvision.transforms": transforms})


_install_stand_in()
import loss as ref_loss      # reference src/loss.py
import rinet                 # reference src/rinet.py

assert len(_Vgg16().features) == 31


def spec_of(mod):
    return [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]


def sample(t, n=16):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].numpy().astype(np.float32)


res = {}
arrs = {}

# ------------------------------------------------------------------ normalize_, window (loss.py:20-32)
res["normalize"] = {"in": [-1024.0, -160.0, 0.0, 240.0, 3072.0],
                    "out": [float(v) for v in ref_loss.normalize_(torch.tensor([-1024.0, -160.0, 0.0, 240.0, 3072.0]))]}
res["window_1d"] = [float(v) for v in ref_loss.gaussian(11, 1.5)]
res["window_2d_row5"] = [float(v) for v in ref_loss.create_window(11, 1)[0, 0, 5]]


# ------------------------------------------------------------------ compute_SSIM / SSIM (loss.py:42-67, 112-130)
def ssim_inputs(case):
    """The input regimes: every test regenerates them from this recipe (kept in loss.json)."""
    kind = case["kind"]
    if kind == "synth":          # HU-like target and its noisy version (raw or clamped to the display window)
        x, y = synth_batch(case["batch"], case["size"], seed=case["seed"])
        if case.get("clamp"):
            x, y = x.clamp(-160.0, 240.0), y.clamp(-160.0, 240.0)
        return y, x              # (target, pred)
    r = np.random.default_rng(case["seed"])       # "flat": nearly constant region, E[x^2] - mu^2 cancels
    shape = (case["batch"], 1, case["size"], case["size"])
    y = case["base"] + case["sd"] * r.standard_normal(shape)
    p = y + case["sd"] * r.standard_normal(shape)
    return torch.from_numpy(y.astype(np.float32)), torch.from_numpy(p.astype(np.float32))


SSIM_CASES = {
    "raw_b4_64": {"kind": "synth", "batch": 4, "size": 64, "seed": 31},
    "clamped_b2_64": {"kind": "synth", "batch": 2, "size": 64, "seed": 32, "clamp": True},
    "raw_b3_128": {"kind": "synth", "batch": 3, "size": 128, "seed": 33},
    "flat_m1000_sd0.5": {"kind": "flat", "batch": 2, "size": 64, "seed": 34, "base": -1000.0, "sd": 0.5},
    "flat_m1000_sd5": {"kind": "flat", "batch": 2, "size": 64, "seed": 35, "base": -1000.0, "sd": 5.0},
    "flat_40_sd20": {"kind": "flat", "batch": 2, "size": 64, "seed": 36, "base": 40.0, "sd": 20.0},
}
for name, case in SSIM_CASES.items():
    t, p = ssim_inputs(case)
    pr = p.clone().requires_grad_(True)
    crit = ref_loss.SSIM()
    d = crit(t, pr)
    d.backward()
    e = dict(case)
    e["ssim_mean"] = float(ref_loss.compute_SSIM(t, p, 11, 1, True))
    e["ssim_per_image"] = [float(v) for v in ref_loss.compute_SSIM(t, p, 11, 1, False)]
    e["dssim"] = float(d)
    # the same formula evaluated in fp64 by the same reference function: how far fp32 rounding alone moves the
    # value in this regime (the tolerance a different fp32 summation order can claim)
    e["ssim_mean_fp64"] = float(ref_loss.compute_SSIM(t.double(), p.double(), 11, 1, True))
    e["dssim_grad_norm"] = float(pr.grad.double().norm())
    p64 = p.double().requires_grad_(True)
    crit(t.double(), p64).backward()               # the same reference code evaluated in fp64
    e["dssim_grad64_norm"] = float(p64.grad.norm())
    res.setdefault("ssim", {})[name] = e
    arrs["dssim_grad:" + name] = pr.grad.numpy()
    arrs["dssim_grad64:" + name] = p64.grad.numpy().astype(np.float32)

# ------------------------------------------------------------------ DLoss / GLoss (loss.py:132-155)
g = torch.Generator().manual_seed(77)
hinge = []
for shift_pos, shift_neg in [(0.3, -0.2), (1.7, -0.2), (0.3, -1.9), (2.5, -3.0), (-0.5, 0.8)]:
    pos = (torch.randn(4, 8192, generator=g) + shift_pos).requires_grad_(True)
    neg = (torch.randn(4, 8192, generator=g) + shift_neg).requires_grad_(True)
    dl = ref_loss.DLoss()(pos, neg)
    dl.backward()
    gl = ref_loss.GLoss()(neg.detach())
    hinge.append({"mean_pos": float(pos.mean()), "mean_neg": float(neg.mean()), "dloss": float(dl),
                  "gloss": float(gl), "dpos": float(pos.grad[0, 0]), "dneg": float(neg.grad[0, 0])})
res["hinge"] = hinge

# ------------------------------------------------------------------ CompoundLoss, MPL, Vgg16FeatureExtractor
vgg_sd = synth_vgg()


def load_vgg(getter):
    getter.load_state_dict(vgg_sd)


crit = ref_loss.CompoundLoss()
load_vgg(crit.model.model)
assert len(crit.model.model) == 30
B, H = 2, 32
pred0, y0 = synth_batch(B, H, seed=41)
d_out0 = torch.randn(B, 64, generator=torch.Generator().manual_seed(3))
pr, dr = pred0.clone().requires_grad_(True), d_out0.clone().requires_grad_(True)
total = crit(pr, y0, dr)
total.backward()
with torch.no_grad():
    fp = crit.model(torch.cat([ref_loss.normalize_(pred0)] * 3, 1))
    fg = crit.model(torch.cat([ref_loss.normalize_(y0)] * 3, 1))
    per_tap = [float(nn.functional.mse_loss(a, b)) for a, b in zip(fp, fg)]
    dss = float(ref_loss.SSIM()(y0, pred0))
    gen = float(ref_loss.GLoss()(d_out0))
res["compound"] = {"batch": B, "size": H, "seed": 41, "d_seed": 3, "loss": float(total), "tap_mse": per_tap,
                   "vgg": float(np.mean(per_tap)), "dssim": dss, "gen": gen,
                   "tap_shapes": [list(t.shape) for t in fp],
                   "dpred_norm": float(pr.grad.double().norm()), "dd_out": float(dr.grad[0, 0])}
arrs["compound:dpred"] = pr.grad.numpy()
for i, t in enumerate(fp):
    arrs[f"compound:tap{i}"] = sample(t, 256)

mpl = ref_loss.MPL()
mpl.model.features.load_state_dict({k: v for k, v in vgg_sd.items()}, strict=False)
pred1, y1 = synth_batch(B, H, seed=43)
pr = pred1.clone().requires_grad_(True)
ml = mpl(y1, pr)
ml.backward()
with torch.no_grad():
    tap8 = ref_loss.get_feature_layer_vgg16(torch.cat([pred1, y1], 0), 8, mpl.model)
res["mpl"] = {"batch": B, "size": H, "seed": 43, "loss": float(ml), "dpred_norm": float(pr.grad.double().norm()),
              "tap8_shape": list(tap8.shape)}
arrs["mpl:dpred"] = pr.grad.numpy()
arrs["mpl:tap8"] = sample(tap8, 256)

json.dump(res, open(os.path.join(HERE, "loss.json"), "w"), indent=1)
np.savez_compressed(os.path.join(HERE, "loss_grads.npz"), **arrs)
print("wrote loss.json, loss_grads.npz", sum(a.nbytes for a in arrs.values()) // 1024, "KiB")

# ------------------------------------------------------------------ training step (train.py:153-177)
G_SEED, D_SEED, X_SEED, NSTEPS = 101, 202, 1234, 3
torch.manual_seed(0)
Gnet, Dnet = rinet.RIGAN(), rinet.ImageDiscriminator()
g0 = synth_state(spec_of(Gnet), G_SEED)
d0 = synth_state(spec_of(Dnet), D_SEED)
Gnet.load_state_dict(g0)
Dnet.load_state_dict(d0)
optG = torch.optim.Adam(Gnet.parameters(), lr=1e-4, betas=(0.01, 0.99))          # train.py:104
optD = torch.optim.Adam(Dnet.parameters(), lr=4e-4, betas=(0.01, 0.99))          # train.py:105
Dloss = ref_loss.DLoss()                                                          # train.py:111
criterion = ref_loss.CompoundLoss()                                               # train.py:112
load_vgg(criterion.model.model)
for p_ in criterion.model.parameters():
    p_.requires_grad_(False)        # the script leaves them trainable and discards the gradients: same numbers
x, y = synth_batch(4, 64, seed=X_SEED)
Dnet.train()
Gnet.train()
step = {}
losses = []
for it in range(NSTEPS):
    pred = Gnet(x)                                                                # train.py:153
    dls = []
    for k in range(2):                                                            # train.py:156
        optD.zero_grad()
        Dnet.zero_grad()
        pred_pos_neg = Dnet(torch.cat([y, pred], dim=0))                          # train.py:160-161
        pred_pos, pred_neg = torch.chunk(pred_pos_neg, 2, dim=0)
        dloss = Dloss(pred_pos, pred_neg).clone()
        dloss.backward(retain_graph=True)                                         # train.py:164
        if k == 1:
            # train.py:172 evaluates the criterion on this pred_neg after optimizer_discriminator.step(); its
            # value does not depend on that order, its gradient is only defined before the in-place update
            gloss = criterion(pred, y, pred_neg)
            gparams = list(Gnet.parameters())
            ggrads = torch.autograd.grad(gloss, gparams + [pred], retain_graph=False)
            dpred, ggrads = ggrads[-1], ggrads[:-1]
            if it == 0:
                with torch.no_grad():
                    fp_ = criterion.model(torch.cat([ref_loss.normalize_(pred)] * 3, 1))
                    fg_ = criterion.model(torch.cat([ref_loss.normalize_(y)] * 3, 1))
                    step["vgg"] = np.array([np.mean([float(nn.functional.mse_loss(a, b)) for a, b in zip(fp_, fg_)])])
                    step["dssim"] = np.array([float(ref_loss.SSIM()(y, pred))])
                    step["gen"] = np.array([float(ref_loss.GLoss()(pred_neg))])
                    step["pred"] = pred.detach().numpy().copy()
                    step["dpred"] = dpred.numpy().copy()
                    step["d_out"] = sample(pred_pos_neg, 512)
                for (n, p_), gr in zip(Gnet.named_parameters(), ggrads):
                    step["ggn:" + n] = np.array([gr.double().norm().item()])
                    step["ggs:" + n] = sample(gr)
        if it == 0:
            for n, p_ in Dnet.named_parameters():
                step[f"dgn{k}:" + n] = np.array([p_.grad.double().norm().item()])
                step[f"dgs{k}:" + n] = sample(p_.grad)
        optD.step()                                                               # train.py:165
        dls.append(float(dloss))
    optG.zero_grad()
    Gnet.zero_grad()                                                              # train.py:169-170
    for p_, gr in zip(gparams, ggrads):
        p_.grad = gr.clone()
    optG.step()                                                                   # train.py:177
    losses.append([dls[0], dls[1], float(gloss)])
    if it == 0:
        for n, v in Gnet.state_dict().items():
            if v.dtype.is_floating_point:
                step["gdn:" + n] = np.array([(v.double() - g0[n].double()).norm().item()])
                step["gds:" + n] = sample(v - g0[n])
        for n, v in Dnet.state_dict().items():
            if v.dtype.is_floating_point:
                step["ddn:" + n] = np.array([(v.double() - d0[n].double()).norm().item()])
                step["dds:" + n] = sample(v - d0[n])
step["losses"] = np.array(losses)            # [step][dloss1, dloss2, gloss]
for n, v in Gnet.state_dict().items():
    if v.dtype.is_floating_point:
        step["gfinal:" + n] = np.array([v.double().norm().item()])
for n, v in Dnet.state_dict().items():
    if v.dtype.is_floating_point:
        step["dfinal:" + n] = np.array([v.double().norm().item()])
np.savez_compressed(os.path.join(HERE, "step_64.npz"), **step)
print("wrote step_64.npz", sum(a.nbytes for a in step.values()) // 1024, "KiB")
print("losses per step [dloss1, dloss2, gloss]:", losses)
