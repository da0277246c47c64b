"""Helpers shared by the GPU parity tests (not collected: no test_ prefix)."""
import torch

from oracle import ried_oracle as O
from reid_gan_amd import engine
from reid_gan_amd.synth import synth_state

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


def weights_agree(a, b, steps=2):
    """Two runs of the same `steps` steps.  The few float atomics left in the step (loss sums) differ in the last bit
    between runs, and Adam's first updates are lr * sign(gradient)-like: a weight whose gradient is almost zero can
    move the other way (up to 2 * lr per step apart: lr 1e-4 for G, 4e-4 for D).  So: nearly all weights agree
    tightly, and none is further apart than those sign flips allow."""
    for st_a, st_b, lr in ((a.gs, b.gs, 1e-4), (a.ds, b.ds, 4e-4)):
        d = (st_a.flat - st_b.flat).abs()
        assert (d > 1.5 * steps * lr).float().mean().item() <= 1e-4, (lr, d.max().item())
        assert d.max().item() <= 2 * steps * lr * 1.05, (lr, d.max().item())


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
