"""Block-level parity on the GPU: every block of src/rinet.py as a HIP plan, forward and backward, against the
golden vectors made from the reference (fp32) and against the bf16-storage-emulating oracle.

Tolerances: output and input gradient vs emulation: mean |err| <= 3e-2 * mean|ref|, max <= 1e-1 * max|ref|;
vs the fp32 golden: output mean |err| <= 5e-2 * mean|ref|, input gradient cosine >= 0.97; parameter gradients vs
emulation: cosine >= 0.985, norm within 6 %.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ried_oracle as O
from reid_gan_amd import engine, ops
from reid_gan_amd.engine import Ctx, Slot, State, Seq
from reid_gan_amd.synth import synth_state

DEV = "cuda"


def _chain1(ctx, xin, out, spec, deconv=False):
    _, cin, cout, k, s, p = spec
    return Seq([engine.ConvBN(ctx, "", cin, cout, k, s, p, xin, out, deconv=deconv)])


# name -> (oracle fn(P, x), plan builder(ctx, xin, out), cin, hw, (cout, hw_out))
BLOCKS = {
    "basic_conv_3x3s2": (lambda P, x: O.basic_conv(P, "", x, 32, 64, 3, 2, 1),
                         lambda c, i, o: _chain1(c, i, o, ("", 32, 64, 3, 2, 1)), 32, 8, (64, 4)),
    "basic_conv_1x7": (lambda P, x: O.basic_conv(P, "", x, 64, 64, (1, 7), 1, (0, 3)),
                       lambda c, i, o: _chain1(c, i, o, ("", 64, 64, (1, 7), 1, (0, 3))), 64, 8, (64, 8)),
    "basic_deconv_4x4s2": (lambda P, x: O.basic_deconv(P, "", x, 96, 32, 4, 2, 1),
                           lambda c, i, o: _chain1(c, i, o, ("", 96, 32, 4, 2, 1), True), 96, 8, (32, 16)),
    "basic_deconv_7x1": (lambda P, x: O.basic_deconv(P, "", x, 128, 128, (7, 1), 1, (3, 0)),
                         lambda c, i, o: _chain1(c, i, o, ("", 128, 128, (7, 1), 1, (3, 0)), True), 128, 4, (128, 4)),
    "inception_a_64": (lambda P, x: O.inception_a(P, "", x, 64, True),
                       lambda c, i, o: engine.inception_a(c, "", i, o, 64, True), 64, 8, (128, 8)),
    "inception_a_256": (lambda P, x: O.inception_a(P, "", x, 256, True),
                        lambda c, i, o: engine.inception_a(c, "", i, o, 256, True), 256, 8, (128, 8)),
    "inception_b_enc": (lambda P, x: O.inception_b(P, "", x, 128, False),
                        lambda c, i, o: engine.inception_b(c, "", i, o, 128, False), 128, 8, (256, 4)),
    "inception_b_dec": (lambda P, x: O.inception_b(P, "", x, 256, True),
                        lambda c, i, o: engine.inception_b(c, "", i, o, 256, True), 256, 4, (256, 8)),
    "inception_c_256": (lambda P, x: O.inception_c(P, "", x, 256, 64, False),
                        lambda c, i, o: engine.inception_c(c, "", i, o, 256, 64, False), 256, 8, (256, 8)),
    "inception_c_512": (lambda P, x: O.inception_c(P, "", x, 512, 64, True),
                        lambda c, i, o: engine.inception_c(c, "", i, o, 512, 64, True), 512, 8, (256, 8)),
    "inception_d_enc": (lambda P, x: O.inception_d(P, "", x, 256, False),
                        lambda c, i, o: engine.inception_d(c, "", i, o, 256, False), 256, 8, (512, 4)),
    "inception_d_dec": (lambda P, x: O.inception_d(P, "", x, 1024, True),
                        lambda c, i, o: engine.inception_d(c, "", i, o, 1024, True), 1024, 4, (512, 8)),
    "inception_e_512": (lambda P, x: O.inception_e(P, "", x, 512, True),
                        lambda c, i, o: engine.inception_e(c, "", i, o, 512, True), 512, 4, (1024, 4)),
}
ORDER = ["basic_conv_3x3s2", "basic_conv_1x7", "basic_deconv_4x4s2", "basic_deconv_7x1", "inception_a_64",
         "inception_a_256", "inception_b_enc", "inception_b_dec", "inception_c_256", "inception_c_512",
         "inception_d_enc", "inception_d_dec", "inception_e_512"]      # index = seed offset used by make_golden.py


class _Stripped(O.Params):
    def get(self, name, shape, kind="param"):
        return super().get(name.lstrip("."), shape, kind)

    def set(self, name, value):
        super().set(name.lstrip("."), value)


def _close(got, ref, what, mx=1e-1, mean=3e-2):
    e = (got.double() - ref.double()).abs()
    assert e.max().item() <= mx * ref.abs().max().item(), f"{what}: max err {e.max().item():.4g} vs {ref.abs().max().item():.4g}"
    assert e.mean().item() <= mean * ref.abs().mean().item(), f"{what}: mean err {e.mean().item():.4g} vs {ref.abs().mean().item():.4g}"


def _cos(got, ref, what, cos=0.985, ratio=0.06):
    a, b = got.double().flatten(), ref.double().flatten()
    c = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()
    r = (a.norm() / (b.norm() + 1e-30)).item()
    if os.environ.get("RIED_DIAG"):
        print(f"DIAG {what}: cosine {c:.5f} ratio {r:.4f}")
    assert c >= cos and abs(r - 1) <= ratio, f"{what}: cosine {c:.4f}, norm ratio {r:.4f}"


def _run_block(name, N, hw, golden=None):
    fn, build, cin, _, (cout, _) = BLOCKS[name]
    bi = ORDER.index(name)
    spec = _Stripped(spec_mode=True)
    with torch.no_grad():
        y0 = fn(spec, torch.zeros(N, cin, hw, hw))
    ho = y0.shape[-1]
    sd = synth_state([(n, s) for n, s, _ in spec.spec], 300 + bi)
    r = np.random.default_rng(500 + bi)
    xin = torch.from_numpy(r.standard_normal((N, cin, hw, hw)).astype(np.float32)).abs()
    # oracle with bf16 storage emulation (the block input is stored in bf16 too)
    osd = {k: (v.clone().requires_grad_(True) if O.trainable(k) else v.clone()) for k, v in sd.items()}
    O.set_bf16_emulation(True)
    try:
        xr = xin.to(torch.bfloat16).float().requires_grad_(True)
        y = fn(_Stripped(osd), xr)
        gw = torch.from_numpy(r.standard_normal(tuple(y.shape)).astype(np.float32))
        names = [k for k in osd if O.trainable(k)]
        grads = torch.autograd.grad((y * gw.to(torch.bfloat16).float()).sum(), [osd[k] for k in names] + [xr])
    finally:
        O.set_bf16_emulation(False)
    # HIP plan
    st = State({"." + k: v for k, v in sd.items()}, DEV)     # the plan is built with prefix "": keys start with "."
    ctx = Ctx(st, DEV, grads=True, training=True)
    a_in = Slot(ctx.act(N, hw, hw, cin))
    a_out = Slot(ctx.act(N, ho, ho, cout))
    plan = build(ctx, a_in, a_out)
    ctx.finalize()
    if name.startswith("inception"):      # the sibling 1x1 layers share launches: the wiring under test here
        grp = plan.layers[0]
        assert isinstance(grp, engine.InputGradGroup) and len(grp.members) >= 2 and grp.bn_jobs is not None, name
    ops.nchw_to_nhwc(xin.to(DEV), a_in.v)
    ctx.begin_forward()
    plan.forward()
    got_y = a_out.v.torch().float().cpu().permute(0, 3, 1, 2)
    ctx.zero_backward(); st.zero_grad()
    ops.nchw_to_nhwc(gw.to(DEV), a_out.g)
    plan.backward()
    ctx.end_backward()
    torch.cuda.synchronize()
    got_dx = a_in.g.torch().float().cpu().permute(0, 3, 1, 2)
    _close(got_y, y.detach(), name + " y")
    _close(got_dx, grads[-1], name + " dx")
    for k, g in zip(names, grads[:-1]):
        _cos(st.g["." + k].cpu(), g, f"{name} d {k}")
    if golden is not None:
        _close(got_y, torch.from_numpy(golden["y"]), name + " y vs golden", mx=2e-1, mean=5e-2)
        # gradients vs the fp32 reference: ReLU / max-pool decisions flip under bf16 rounding of the input, so
        # individual elements may differ a lot; the direction and the size of the gradient must agree
        _cos(got_dx, torch.from_numpy(golden["dx"]), name + " dx vs golden", cos=0.97, ratio=0.1)


@pytest.mark.parametrize("name", ORDER)
def test_block_against_golden_and_oracle(golden_dir, name):
    _, _, _, hw, _ = BLOCKS[name]
    _run_block(name, 2, hw, np.load(os.path.join(golden_dir, f"block_{name}.npz")))


@pytest.mark.parametrize("name,N,hw", [("basic_deconv_4x4s2", 2, 32), ("basic_conv_3x3s2", 3, 32),
                                       ("inception_a_64", 4, 16), ("inception_b_dec", 3, 8),
                                       ("inception_e_512", 8, 4), ("inception_d_enc", 2, 16)])
def test_block_larger_shapes(name, N, hw):
    """Shapes that make the weight-gradient K loop run several tiles per split and several splits."""
    _run_block(name, N, hw)


# The blocks of the generator at the shapes of the benchmarked training step (batch 128 of 64 x 64 patches: maps of
# 16 x 16, 8 x 8 and 4 x 4 inside the blocks, 32 x 32 for conv2 / deconv2), where the plans take other tiles, split-K
# factors and fold rows than the two-image cases above: same tolerances.
BENCH_SHAPES = [("basic_conv_3x3s2", 32), ("basic_deconv_4x4s2", 32), ("inception_a_64", 16), ("inception_a_256", 16),
                ("inception_b_enc", 16), ("inception_b_dec", 8), ("inception_c_256", 8), ("inception_c_512", 8),
                ("inception_d_enc", 8), ("inception_d_dec", 4), ("inception_e_512", 4)]


def _plan_features(name, N, hw):
    """(Par objects, ConvBN layers) of the block's plan at that shape (no kernels run)."""
    _, build, cin, _, (cout, _) = BLOCKS[name]
    bi = ORDER.index(name)
    spec = _Stripped(spec_mode=True)
    with torch.no_grad():
        ho = BLOCKS[name][0](spec, torch.zeros(1, cin, hw, hw)).shape[-1]
    st = State({"." + k: v for k, v in synth_state([(n, s) for n, s, _ in spec.spec], 300 + bi).items()}, DEV)
    ctx = Ctx(st, DEV, grads=True, training=True)
    plan = build(ctx, Slot(ctx.act(N, hw, hw, cin)), Slot(ctx.act(N, ho, ho, cout)))
    ctx.finalize()
    pars, convs = [], []

    def walk(l):
        if isinstance(l, engine.Seq):
            for q in l.layers:
                walk(q)
        elif isinstance(l, engine.Par):
            pars.append(l)
            for b in l.branches:
                walk(b)
        elif isinstance(l, engine.InputGradGroup):
            for m in l.members:
                walk(m)
        elif isinstance(l, engine.ConvBN):
            convs.append(l)
    walk(plan)
    return pars, convs


@pytest.mark.parametrize("name,hw", BENCH_SHAPES)
def test_block_at_the_benchmark_batch(name, hw):
    N = 128
    if name.startswith("inception"):
        pars, convs = _plan_features(name, N, hw)
        # the code paths of the benchmarked plan are the ones under test: level-wise job tables for the branches,
        # BatchNorm sums folded with atomics forward and backward (no finalize / finish launches)
        assert pars and all(p.levels is not None for p in pars), name
        assert convs and all(c.sliced and c.bwd_fold for c in convs), name
    _run_block(name, N, hw)
