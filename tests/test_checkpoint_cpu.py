"""Checkpoint interchange (src/train.py:77-97,196-205): parameter order and optimiser-state layout against the
structure recorded from the reference's own modules and torch.optim.Adam
(tests/golden/checkpoint_structure.json, made by tests/golden/make_golden_checkpoint.py); file round trips."""
import json
import os

import pytest
import torch

from reid_gan_amd import checkpoint as C
from reid_gan_amd import rinet
from reid_gan_amd.engine import State, is_trainable
from reid_gan_amd.synth import synth_state


@pytest.fixture(scope="module")
def structure(golden_dir):
    return json.load(open(os.path.join(golden_dir, "checkpoint_structure.json")))


@pytest.mark.parametrize("net", ["G", "D"])
def test_parameter_order_is_the_reference_optimisers(structure, net):
    names = [k for k in C.key_order(net) if is_trainable(k)]
    ent = structure[net]["entries"]
    assert names == [e["name"] for e in ent] and len(names) == structure[net]["n_params"]
    m = rinet.RIGAN() if net == "G" else rinet.ImageDiscriminator()
    shapes = {k: list(p.shape) for k, p in m.named_parameters()}
    assert [shapes[e["name"]] for e in ent] == [e["shape"] for e in ent]
    assert list(C.KEYS) == structure["keys"]


def _d_state(seed):
    m = rinet.ImageDiscriminator()
    spec = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    return State(synth_state(spec, seed), "cpu")


def _same(a: State, b: State, attr: str) -> bool:
    """Equal on every parameter's slice of a flat buffer (the alignment padding between tensors is not state)."""
    return all(torch.equal(getattr(a, attr)[o:o + a.p[k].numel()], getattr(b, attr)[o:o + a.p[k].numel()])
               for k, o in a.offs.items())


def test_file_round_trip_and_torch_adam_accepts_it(structure, tmp_path):
    ds, gs = _d_state(5), _d_state(6)            # two discriminator-shaped states stand in for (G, D): small
    for st, n in ((ds, 3), (gs, 0)):
        g = torch.Generator().manual_seed(n)
        if n:
            st.adam_m.copy_(torch.randn(st.adam_m.shape, generator=g))
            st.adam_v.copy_(torch.rand(st.adam_v.shape, generator=g))
            st.adam_step.fill_(n)
    ck = {"epoch": 7, "netG_state_dict": C.net_state_dict(gs, "D"), "optG_state_dict": C.optim_state_dict(gs, "D", 1e-4),
          "netD_state_dict": C.net_state_dict(ds, "D"), "optD_state_dict": C.optim_state_dict(ds, "D", 4e-4),
          "lr": 1e-4, "total_iters": 350}
    assert list(ck) == structure["keys"] and ck["optG_state_dict"]["state"] == {}      # no step taken: no state yet
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
        got, want = ck["optD_state_dict"]["param_groups"][0][k], structure["D"]["group"][k]
        assert (list(got) if isinstance(got, tuple) else got) == want, k
    path = str(tmp_path / "latest_ckpt.pth.tar")
    torch.save(ck, path)
    back = C.load_checkpoint(path)                                # weights_only load
    assert back["epoch"] == 7 and back["total_iters"] == 350
    assert list(back["netD_state_dict"]) == C.key_order("D")      # the reference's key order
    fresh = _d_state(99)
    C.load_net_state(fresh, back["netD_state_dict"])
    assert C.load_optim_state(fresh, "D", back["optD_state_dict"]) == pytest.approx(4e-4)
    assert _same(fresh, ds, "flat") and _same(fresh, ds, "adam_m") and _same(fresh, ds, "adam_v")
    assert fresh.adam_step.item() == 3
    for k in ds.b:
        assert torch.equal(fresh.b[k], ds.b[k]), k
    # the module route: torch's own loaders take the file's entries
    m = rinet.ImageDiscriminator()
    m.load_state_dict(back["netD_state_dict"])
    opt = torch.optim.Adam(m.parameters(), lr=4e-4, betas=(0.01, 0.99))
    opt.load_state_dict(back["optD_state_dict"])
    names = [k for k, _ in m.named_parameters()]
    for i, p in enumerate(m.parameters()):
        o, n = ds.offs[names[i]], p.numel()
        assert torch.equal(opt.state[p]["exp_avg"].reshape(-1), ds.adam_m[o:o + n]), names[i]
        assert float(opt.state[p]["step"]) == 3.0
    # and back: what torch's optimiser saves is what load_optim_state reads
    again = _d_state(98)
    C.load_optim_state(again, "D", opt.state_dict())
    assert _same(again, ds, "adam_v") and again.adam_step.item() == 3


def test_mismatched_files_are_refused(tmp_path):
    ds = _d_state(5)
    sd = C.net_state_dict(ds, "D")
    sd.pop("discriminator_net.3.gamma")
    with pytest.raises(KeyError, match="missing"):
        C.load_net_state(ds, sd)
    osd = C.optim_state_dict(ds, "D", 4e-4)
    osd["param_groups"][0]["params"] = osd["param_groups"][0]["params"][:-1]
    with pytest.raises(ValueError, match="one group"):
        C.load_optim_state(ds, "D", osd)
    torch.save({"something": 1}, str(tmp_path / "x.pt"))
    with pytest.raises(KeyError, match="not a RIED-GAN checkpoint"):
        C.load_checkpoint(str(tmp_path / "x.pt"))
