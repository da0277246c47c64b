"""Checkpoint interchange with the reference's training script (src/train.py:77-97 load, :196-205 save;
src/test.py:109-114 load for inference).

The file is what torch.save writes for the 7-key dict
    {'epoch', 'netG_state_dict', 'optG_state_dict', 'netD_state_dict', 'optD_state_dict', 'lr', 'total_iters'}
with the optimizer entries in torch.optim.Adam.state_dict() form: state[i] = {'step', 'exp_avg', 'exp_avg_sq'} for
the i-th tensor of module.parameters(), one param group.  The drop-in modules (reid_gan_amd.rinet) are nn.Modules
with the reference's parameter order, so they read and write such files through torch itself; this module does the
same for the fused training step (TrainStep), whose parameters and Adam moments live in flat buffers.

Files are read with torch.load(weights_only=True): nothing in them is executed.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .engine import State, is_trainable

KEYS = ("epoch", "netG_state_dict", "optG_state_dict", "netD_state_dict", "optD_state_dict", "lr", "total_iters")
_ORDER: Dict[str, List[str]] = {}


def key_order(net: str) -> List[str]:
    """state_dict keys of RIGAN ('G') / ImageDiscriminator ('D') in the reference's registration order."""
    if net not in _ORDER:
        from . import rinet
        m = rinet.RIGAN() if net == "G" else rinet.ImageDiscriminator()
        _ORDER[net] = list(m.state_dict().keys())
    return _ORDER[net]


def net_state_dict(st: State, net: str) -> Dict[str, torch.Tensor]:
    """The network's state_dict in the reference's key order (CPU copies)."""
    out = {}
    for k in key_order(net):
        out[k] = (st.p[k] if k in st.p else st.b[k]).detach().cpu().clone()
    return out


def optim_state_dict(st: State, net: str, lr: float, betas=(0.01, 0.99), eps=1e-8) -> dict:
    """torch.optim.Adam.state_dict() for the flat Adam buffers of `st` (src/train.py:104-105: betas (0.01, 0.99))."""
    names = [k for k in key_order(net) if is_trainable(k)]
    step = int(st.adam_step.item())
    state = {}
    if step > 0:              # torch creates the per-parameter state lazily, at the first step
        for i, k in enumerate(names):
            o, n, shape = st.offs[k], st.p[k].numel(), st.p[k].shape
            state[i] = {"step": torch.tensor(float(step)),
                        "exp_avg": st.adam_m[o:o + n].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": st.adam_v[o:o + n].view(shape).detach().cpu().clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False,
             "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def load_net_state(st: State, sd: Dict[str, torch.Tensor]):
    """Parameters and buffers from a reference state_dict (every key must be present, shapes must match)."""
    want = set(st.p) | set(st.b)
    missing, extra = want - set(sd), set(sd) - want
    if missing or extra:
        raise KeyError(f"state_dict mismatch: missing {sorted(missing)[:4]}, unexpected {sorted(extra)[:4]}")
    for k, v in sd.items():
        dst = st.p[k] if k in st.p else st.b[k]
        if tuple(dst.shape) != tuple(v.shape):
            raise ValueError(f"{k}: shape {tuple(v.shape)} in the file, {tuple(dst.shape)} expected")
        dst.copy_(v.to(dst.dtype))


def load_optim_state(st: State, net: str, osd: dict) -> float:
    """Adam moments and step count from a torch.optim.Adam.state_dict(); returns the group's learning rate."""
    names = [k for k in key_order(net) if is_trainable(k)]
    groups = osd["param_groups"]
    if len(groups) != 1 or len(groups[0]["params"]) != len(names):
        raise ValueError(f"optimizer state for {len(names)} parameters in one group expected")
    st.adam_m.zero_(); st.adam_v.zero_(); st.adam_step.zero_()
    steps = set()
    for i, k in zip(groups[0]["params"], names):
        s = osd["state"].get(i)
        if s is None:
            continue
        o, n = st.offs[k], st.p[k].numel()
        if s["exp_avg"].numel() != n:
            raise ValueError(f"{k}: optimizer state has {s['exp_avg'].numel()} elements, {n} expected")
        st.adam_m[o:o + n].copy_(s["exp_avg"].reshape(-1))
        st.adam_v[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
        steps.add(int(float(s["step"])))
    if len(steps) > 1:
        raise ValueError(f"parameters at different Adam steps {sorted(steps)}: the fused step keeps one counter")
    if steps:
        st.adam_step.fill_(steps.pop())
    return float(groups[0]["lr"])


def checkpoint_dict(gs: State, ds: State, epoch: int, total_iters: int, lr: float) -> dict:
    """The dict src/train.py:196-204 saves (G at lr, D at 4*lr: train.py:104-105)."""
    return {"epoch": epoch, "netG_state_dict": net_state_dict(gs, "G"),
            "optG_state_dict": optim_state_dict(gs, "G", lr),
            "netD_state_dict": net_state_dict(ds, "D"), "optD_state_dict": optim_state_dict(ds, "D", 4 * lr),
            "lr": lr, "total_iters": total_iters}


def save_checkpoint(path: str, ts, epoch: int, total_iters: int, lr: Optional[float] = None):
    """ts: a TrainStep.  Writes a file the reference's train.py (--load_chkpt) and test.py load."""
    ck = checkpoint_dict(ts.gs, ts.ds, epoch, total_iters, ts.lr if lr is None else lr)
    # A look-ahead plan has already run the next step's spectral-norm power iteration when a step returns: the
    # weight_u / weight_v saved here are one iteration ahead of what the reference's loop would hold at this point.
    # The extra key (ignored by the reference's loaders, which read their seven keys by name) lets restore() resume
    # a look-ahead plan without repeating that iteration.
    if getattr(ts, "lookahead", False) and getattr(ts, "_d_ready", False):
        ck["ried_d_prepared"] = True
    torch.save(ck, path)


def load_checkpoint(path: str) -> dict:
    ck = torch.load(path, map_location="cpu", weights_only=True)
    missing = [k for k in ("netG_state_dict",) if k not in ck]
    if missing:
        raise KeyError(f"{path}: not a RIED-GAN checkpoint (no {missing[0]})")
    return ck


def restore(ts, ck: dict, optimizers: bool = True) -> dict:
    """Put a loaded checkpoint into a TrainStep (networks, and Adam state when present and wanted).  Returns
    {'epoch', 'total_iters', 'lr'} as stored.  With the optimizer state comes the learning rate the optimizers
    were saved with (param_groups[0]['lr'] of optG_state_dict: torch's load_state_dict restores it the same way,
    train.py:91-92); the top-level 'lr' entry is only reported, as in the reference (train.py:94,97)."""
    load_net_state(ts.gs, ck["netG_state_dict"])
    if "netD_state_dict" in ck:
        load_net_state(ts.ds, ck["netD_state_dict"])
    if optimizers and "optG_state_dict" in ck:
        load_optim_state(ts.gs, "G", ck["optG_state_dict"])
        load_optim_state(ts.ds, "D", ck["optD_state_dict"])
        groups = ck["optG_state_dict"].get("param_groups") or []
        if groups and "lr" in groups[0]:
            ts.lr = float(groups[0]["lr"])
    if hasattr(ts, "weights_changed"):      # a look-ahead plan has prepared its next step with the old weights
        ts.weights_changed(d_prepared=bool(ck.get("ried_d_prepared", False)))
    return {k: ck.get(k) for k in ("epoch", "total_iters", "lr")}
