"""Generate tests/golden/legacy_*.npz by importing the reference's src/networks.py (GNet, DNet).

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_legacy.py

The module is imported unchanged, loaded with seeded synthetic state dicts (reid_gan_amd.synth) and run on seeded
inputs; only numbers are stored.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch

import networks         # reference
from reid_gan_amd.synth import synth_state, synth_batch

torch.set_num_threads(8)
GNET_SEED, DNET_SEED = 404, 505


def spec_of(mod):
    return [(k, tuple(v.shape)) for k, v in mod.state_dict().items()]


def sample(t, n=512):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].numpy().astype(np.float32)


def stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.std().item()], dtype=np.float64)


def grad_summary(mod):
    out = {}
    for k, p in mod.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out["gn:" + k] = np.array([g.double().norm().item()])
        out["gs:" + k] = sample(g, 16)
    return out


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("wrote", name, sum(a.nbytes for a in arrs.values()) // 1024, "KiB")


def run(mod, seed, x, tops, running):
    mod.load_state_dict(synth_state(spec_of(mod), seed))
    mod.train()
    x = x.clone().requires_grad_(True)
    acts = {}
    hooks = [getattr(mod, n).register_forward_hook(lambda m, i, o, n=n: acts.__setitem__(n, o.detach().clone()))
             for n in tops]
    out = mod(x)
    for h in hooks:
        h.remove()
    r = np.random.default_rng(seed + 1)
    gw = torch.from_numpy(r.standard_normal(tuple(out.shape)).astype(np.float32))
    (out * gw).sum().backward()
    arrs = {"out": out.detach().numpy(), "dx": x.grad.numpy()}
    for n in tops:                 # a hook on the shared residual block keeps its last (fifth) application
        arrs["st:" + n] = stats(acts[n])
        arrs["sa:" + n] = sample(acts[n])
    sd = mod.state_dict()
    for k in running:
        arrs["rs:" + k] = sd[k].numpy()
    arrs.update(grad_summary(mod))
    return arrs


torch.manual_seed(0)
x, _ = synth_batch(2, 64, seed=41)
save("legacy_gnet_train_64.npz",
     **run(networks.GNet(), GNET_SEED, x, ["conv1", "layer1", "layer2", "residual", "deconv1", "deconv2"],
           ["layer1.1.running_mean", "layer1.1.running_var", "residual.1.running_mean", "residual.1.running_var",
            "residual.4.running_mean", "residual.1.num_batches_tracked", "deconv2.1.running_mean"]))

x, _ = synth_batch(4, 64, seed=42)
save("legacy_dnet_train_64.npz",
     **run(networks.DNet(), DNET_SEED, x, ["conv1", "conv2", "conv3", "conv4", "conv5"],
           ["conv2.1.running_mean", "conv2.1.running_var", "conv5.1.running_mean", "conv5.1.num_batches_tracked"]))

# eval-mode forwards (running statistics; the conv biases matter here)
G = networks.GNet(); G.load_state_dict(synth_state(spec_of(G), GNET_SEED)); G.eval()
D = networks.DNet(); D.load_state_dict(synth_state(spec_of(D), DNET_SEED)); D.eval()
x, _ = synth_batch(2, 64, seed=43)
with torch.no_grad():
    save("legacy_eval_64.npz", gnet=G(x).numpy(), dnet=D(x).numpy())
