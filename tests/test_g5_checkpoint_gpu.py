"""One Adam step of the flat-buffer optimiser, exported as a checkpoint, against the same step taken by the
reference's modules with torch.optim.Adam (tests/golden/checkpoint_structure.json): pins the kernel's arithmetic
and the parameter order at once.  fp32: relative tolerance 1e-5."""
import json
import os

import pytest
import torch

from reid_gan_amd import checkpoint as C
from reid_gan_amd import rinet
from reid_gan_amd.engine import State
from reid_gan_amd.synth import synth_state, synth_tensor, synth_batch, synth_vgg
from reid_gan_amd.train_step import TrainStep

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _spec(net):
    m = rinet.RIGAN() if net == "G" else rinet.ImageDiscriminator()
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


@pytest.mark.parametrize("net", ["G", "D"])
def test_adam_step_matches_reference_optimiser(golden_dir, net):
    S = json.load(open(os.path.join(golden_dir, "checkpoint_structure.json")))[net]
    st = State(synth_state(_spec(net), S["seed"]), DEV)
    for k in st.g:
        st.g[k].copy_(synth_tensor("grad:" + k, tuple(st.g[k].shape), S["seed"] + 1) * 1e-2)
    st.adam(S["lr"])
    osd = C.optim_state_dict(st, net, S["lr"])
    sd = C.net_state_dict(st, net)
    for i, e in enumerate(S["entries"]):
        s = osd["state"][i]
        assert float(s["step"]) == e["step"] == 1.0 and list(s["exp_avg"].shape) == e["shape"]
        for key, got in (("exp_avg", s["exp_avg"]), ("exp_avg_sq", s["exp_avg_sq"]), ("param_after", sd[e["name"]])):
            g = got.reshape(-1)[:3].double()
            w = torch.tensor(e[key], dtype=torch.float64)
            assert torch.allclose(g, w, rtol=1e-5, atol=1e-12), (e["name"], key, g, w)


def test_train_step_checkpoint_round_trip(tmp_path):
    """Save after two steps, restore into a fresh TrainStep, take one more step in both: same state."""
    B, H = 4, 64
    gsd, dsd, vgg = synth_state(_spec("G"), 101), synth_state(_spec("D"), 202), synth_vgg()
    x, y = synth_batch(B, H, seed=3)
    a = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
    a.step(x.to(DEV), y.to(DEV)); a.step()
    path = str(tmp_path / "latest_ckpt.pth.tar")
    C.save_checkpoint(path, a, epoch=1, total_iters=2)
    b = TrainStep(synth_state(_spec("G"), 7), synth_state(_spec("D"), 8), vgg, B, H, H, lr=1e-4, use_graph=True)
    meta = C.restore(b, C.load_checkpoint(path))
    assert meta == {"epoch": 1, "total_iters": 2, "lr": 1e-4}
    assert torch.equal(b.gs.flat, a.gs.flat) and torch.equal(b.ds.adam_v, a.ds.adam_v)
    assert b.gs.adam_step.item() == 2 and b.ds.adam_step.item() == 4       # D steps twice per iteration
    a.step(); b.step(x.to(DEV), y.to(DEV))                                  # b captures its graph here
    torch.cuda.synchronize()
    assert b.gs.adam_step.item() == 3 and b.ds.adam_step.item() == 6
    la, lb = a.losses(), b.losses()
    for k in la:
        assert abs(la[k] - lb[k]) <= 5e-3 * abs(la[k]) + 1e-6, (k, la[k], lb[k])
    assert torch.allclose(a.gs.flat, b.gs.flat, rtol=0, atol=2.5e-4)       # one Adam step apart at most: lr = 1e-4


def test_restored_learning_rate_is_applied_and_recaptures(tmp_path):
    """A checkpoint saved at another learning rate: restore() applies the optimizer's stored rate, the captured
    graphs (which hold the rate as a launch argument) are dropped, and the next step moves the parameters by the
    new rate: as far as the twin that has been stepping at that rate all along."""
    B, H = 2, 64
    gsd, dsd, vgg = synth_state(_spec("G"), 101), synth_state(_spec("D"), 202), synth_vgg()
    x, y = synth_batch(B, H, seed=3)
    a = TrainStep(gsd, dsd, vgg, B, H, H, lr=4e-4, use_graph=False)
    a.step(x.to(DEV), y.to(DEV))
    path = str(tmp_path / "ckpt.pth.tar")
    C.save_checkpoint(path, a, epoch=0, total_iters=1)
    b = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=True)
    b.step(x.to(DEV), y.to(DEV))
    assert b.graphs is not None and b.lr == 1e-4
    C.restore(b, C.load_checkpoint(path))
    assert b.lr == 4e-4 and b.graphs is None
    assert torch.equal(a.gs.flat, b.gs.flat)
    before = b.gs.flat.clone()
    a.step(); b.step()
    torch.cuda.synchronize()
    assert b.graphs is not None
    move_a = (a.gs.flat - before).abs().median().item()
    move_b = (b.gs.flat - before).abs().median().item()
    assert 0.8 < move_b / move_a < 1.25, (move_a, move_b)      # a step at the old rate would give 0.25


def test_restore_into_a_lookahead_plan(tmp_path):
    """A look-ahead plan has prepared its next step (target features, the discriminator's packed weights and its
    pass over the targets) with the weights it had: restore() tells it, and the step after the restore is the step
    the plain plan takes from the same checkpoint."""
    B, H = 4, 64
    gsd, dsd, vgg = synth_state(_spec("G"), 101), synth_state(_spec("D"), 202), synth_vgg()
    x, y = (t.to(DEV) for t in synth_batch(B, H, seed=3))
    a = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
    a.step(x, y); a.step()
    path = str(tmp_path / "ckpt.pth.tar")
    C.save_checkpoint(path, a, epoch=1, total_iters=2)
    b = TrainStep(synth_state(_spec("G"), 7), synth_state(_spec("D"), 8), vgg, B, H, H, lr=1e-4, use_graph=True,
                  lookahead=True)
    b.step(x, y)                       # captures; ends with the next step prepared from b's own weights
    assert b._primed and b._d_ready
    C.restore(b, C.load_checkpoint(path))
    assert not b._primed and not b._d_ready
    a.step(); b.step(x, y)
    torch.cuda.synchronize()
    la, lb = a.losses(), b.losses()
    for k in la:
        assert abs(la[k] - lb[k]) <= 5e-3 * abs(la[k]) + 1e-6, (k, la[k], lb[k])
    assert torch.allclose(a.gs.flat, b.gs.flat, rtol=0, atol=2.5e-4)


def test_lookahead_plan_resumes_without_a_second_power_iteration(tmp_path, monkeypatch):
    """A checkpoint written after a look-ahead step holds spectral-norm vectors that already include the next step's
    power iteration; the file says so (key `ried_d_prepared`, beside the reference's seven keys) and a look-ahead
    plan restored from it rebuilds sigma and the packed weights from those vectors without iterating again: the
    resumed run takes the same next step as the uninterrupted one."""
    from reid_gan_amd import engine
    monkeypatch.setattr(engine, "BN_FOLD", False)        # fixed-order reductions: the runs must agree closely
    B, H = 4, 64
    gsd, dsd, vgg = synth_state(_spec("G"), 101), synth_state(_spec("D"), 202), synth_vgg()
    x, y = (t.to(DEV) for t in synth_batch(B, H, seed=3))
    a = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False, lookahead=True)
    a.step(x, y)
    path = str(tmp_path / "ckpt.pth.tar")
    C.save_checkpoint(path, a, epoch=1, total_iters=1)
    ck = C.load_checkpoint(path)
    assert ck.get("ried_d_prepared") is True
    b = TrainStep(synth_state(_spec("G"), 7), synth_state(_spec("D"), 8), vgg, B, H, H, lr=1e-4, use_graph=False,
                  lookahead=True)
    C.restore(b, ck)
    assert b._resume_prepared and not b._d_ready
    # ... and a third plan restored from the same file with the key taken away: it iterates once more
    c = TrainStep(synth_state(_spec("G"), 7), synth_state(_spec("D"), 8), vgg, B, H, H, lr=1e-4, use_graph=False,
                  lookahead=True)
    C.restore(c, {k: v for k, v in ck.items() if k != "ried_d_prepared"})
    assert not c._resume_prepared
    a.step(x, y); b.step(x, y); c.step(x, y)
    torch.cuda.synchronize()
    # Same number of power iterations: the vectors agree up to the round-off of two runs whose reductions are not
    # bit-reproducible (float atomics in the split weight gradients), 2e-3 absolute on unit vectors, and are at
    # least five times closer than the run that iterated once more.
    for k in ("discriminator_net.0.conv2d.weight_u", "discriminator_net.6.conv2d.weight_v"):
        d_ab = (a.ds.b[k] - b.ds.b[k]).abs().max().item()
        d_ac = (a.ds.b[k] - c.ds.b[k]).abs().max().item()
        assert d_ab <= 2e-3 and 5 * d_ab < d_ac, (k, d_ab, d_ac)
    # Losses: what is computed before the step's first Adam update agrees tightly; dloss2 and the adversarial term
    # come after the discriminator's first update, which moves every weight by about lr whatever the size of its
    # gradient (Adam's first steps), so round-off in a gradient's sign shows there: the tolerance of the other
    # restore tests.
    la, lb = a.losses(), b.losses()
    for k in la:
        tol = 1e-2 if k in ("dloss2", "gen", "gloss") else 1e-3
        assert abs(la[k] - lb[k]) <= tol * abs(la[k]) + 1e-6, (k, la[k], lb[k])
    # a plain plan written from has no such key
    p = TrainStep(gsd, dsd, vgg, B, H, H, lr=1e-4, use_graph=False)
    p.step(x, y)
    C.save_checkpoint(path, p, epoch=1, total_iters=1)
    assert "ried_d_prepared" not in C.load_checkpoint(path)
