"""The oracle's GNet / DNet (src/networks.py:460-556) against the golden vectors made by importing the reference
(tests/golden/make_golden_legacy.py), and the drop-in modules' parameter names."""
import os

import numpy as np
import torch

from oracle import ried_oracle as O
from reid_gan_amd.synth import synth_state, synth_batch

GNET_SEED, DNET_SEED = 404, 505


def _sample(t, n=512):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride][:n].numpy()


def _close(a, b, rtol=3e-4, atol=3e-5, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= atol * max(scale, 1.0) + rtol * scale, f"{what}: max err {err} vs scale {scale}"


def _run(spec, fwd, seed, x, g, tops):
    sd = synth_state([(n, s) for n, s, _ in spec], seed)
    sd = {k: (v.requires_grad_(True) if O.trainable(k) else v) for k, v in sd.items()}
    P = O.Params(sd)
    x = x.clone().requires_grad_(True)
    acts = {}
    out = fwd(P, x, True, acts)
    _close(out.detach().numpy(), g["out"], what="out")
    for n in tops:
        _close(_sample(acts[n]), g["sa:" + n], what=n)
    gw = torch.from_numpy(np.random.default_rng(seed + 1).standard_normal(tuple(out.shape)).astype(np.float32))
    names = [k for k in sd if O.trainable(k)]
    grads = torch.autograd.grad((out * gw).sum(), [sd[k] for k in names] + [x])
    _close(grads[-1].numpy(), g["dx"], rtol=3e-3, what="dx")
    for k, gr in zip(names, grads[:-1]):
        ref = g["gn:" + k][0]
        # the bias of a conv in front of a train-mode BatchNorm has a gradient of rounding noise only
        tol = 3e-3 * ref + 1e-9 if not (k.endswith(".bias") and ref < 1e-3) else 1e-3
        assert abs(gr.double().norm().item() - ref) <= tol, (k, gr.norm().item(), ref)
    for k in g.files:
        if k.startswith("rs:"):
            _close(P.sd[k[3:]].detach().numpy(), g[k], what=k)


def test_gnet_oracle_against_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "legacy_gnet_train_64.npz"))
    x, _ = synth_batch(2, 64, seed=41)
    _run(O.gnet_spec(), O.gnet_forward, GNET_SEED, x, g, ["conv1", "layer1", "layer2", "residual", "deconv1", "deconv2"])
    assert int(g["rs:residual.1.num_batches_tracked"]) == 5          # one block object, five applications


def test_dnet_oracle_against_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "legacy_dnet_train_64.npz"))
    x, _ = synth_batch(4, 64, seed=42)
    _run(O.dnet_spec(), O.dnet_forward, DNET_SEED, x, g, ["conv1", "conv2", "conv3", "conv4", "conv5"])


def test_legacy_eval_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "legacy_eval_64.npz"))
    x, _ = synth_batch(2, 64, seed=43)
    with torch.no_grad():
        sd = synth_state([(n, s) for n, s, _ in O.gnet_spec()], GNET_SEED)
        _close(O.gnet_forward(O.Params(sd), x, False).numpy(), g["gnet"], what="gnet eval")
        sd = synth_state([(n, s) for n, s, _ in O.dnet_spec()], DNET_SEED)
        _close(O.dnet_forward(O.Params(sd), x, False).numpy(), g["dnet"], what="dnet eval")


def test_bf16_emulation_drops_the_bias_consistently():
    """With the emulation on the conv in front of a train-mode BatchNorm runs without bias (as the HIP path does);
    outputs and running statistics must stay those of the plain arithmetic up to bf16 rounding."""
    x, _ = synth_batch(4, 64, seed=44)
    outs, rms = [], []
    for emu in (False, True):
        O.set_bf16_emulation(emu)
        try:
            sd = synth_state([(n, s) for n, s, _ in O.dnet_spec()], DNET_SEED)
            P = O.Params(sd)
            with torch.no_grad():
                outs.append(O.dnet_forward(P, x, True))
            rms.append(P.sd["conv3.1.running_mean"].clone())
        finally:
            O.set_bf16_emulation(False)
    # conv5 normalises 16 values per channel: single outputs move with bf16 rounding, the bulk must not
    assert (outs[0] - outs[1]).abs().mean() < 0.01
    assert (rms[0] - rms[1]).abs().max() < 0.02 * rms[0].abs().max() + 1e-3


def test_module_parameter_names_match_reference_layout():
    """The drop-in modules hold the keys of the reference's state_dict (same names and shapes as the oracle's
    spec, which the goldens above pin to the reference)."""
    from reid_gan_amd import networks as N
    for cls, spec in ((N.GNet, O.gnet_spec()), (N.DNet, O.dnet_spec())):
        got = {k: tuple(v.shape) for k, v in cls().state_dict().items()}
        want = {n: tuple(s) for n, s, _ in spec}
        assert got == want
