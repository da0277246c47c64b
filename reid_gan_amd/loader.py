"""Training-patch loader with the reference's names (src/loader.py): ct_dataset, get_patch, get_loader.

The reference keeps the int16 HU slices in host memory (load_mode 1) or on disk (load_mode 0), cuts patches with
numpy in DataLoader worker processes and ships them to the GPU every step.  Here the slices of the chosen split
are loaded once and stay resident in HBM as two int16 tensors [S][H][W]; every batch is one launch of
ried_patch_crop that crops and converts on the device.  What a caller of the reference's loader sees is unchanged:

  for x, y in get_loader('train', 1, saved_path, 'L064', patch_n, patch_size, None, batch_size):
      # x, y: int16 (batch_size, patch_n, P, P) device tensors - train.py:136-142 turns them into fp32 (-1,1,P,P)

and `loader.float_batches()` yields that fp32 (batch_size*patch_n, 1, P, P) form directly (conversion fused into
the crop kernel).  Crop positions: rng="numpy" (default) draws them on the host with numpy.random.randint in the
reference's order (top, then left, per patch, per slice: src/loader.py:66-68), so a seeded numpy generator gives
the reference's patches bit for bit; rng="device" draws them inside the kernel from (seed, running draw count)
and involves the host in nothing but the launch.
"""
from __future__ import annotations

import os
from glob import glob
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from ._lib import lib
from .ops import _p, _stream


def _crop(slices_in, slices_tg, first_slice, nsel, patch_n, P, offsets, seed, first_draw, out_f32, offsets_out=None):
    S, H, W = slices_in.shape
    n = nsel * patch_n
    dt = torch.float32 if out_f32 else torch.int16
    x = torch.empty(n, P, P, dtype=dt, device=slices_in.device)
    y = torch.empty(n, P, P, dtype=dt, device=slices_in.device)
    lib.ried_patch_crop(_p(slices_in), _p(slices_tg), S, H, W, first_slice, nsel, patch_n, P, _p(offsets), seed,
                        first_draw, _p(x), _p(y), int(out_f32), _p(offsets_out), _stream())
    return x, y


def _host_offsets(n: int, h: int, w: int, P: int) -> np.ndarray:
    """(top, left) per patch from numpy's global generator, in the order of src/loader.py:66-68."""
    out = np.empty((n, 2), dtype=np.int32)
    for j in range(n):
        out[j, 0] = np.random.randint(0, h - P)
        out[j, 1] = np.random.randint(0, w - P)
    return out


def get_patch(full_input_img: torch.Tensor, full_target_img: torch.Tensor, patch_n: int, patch_size: int):
    """src/loader.py:60-73 on device tensors: (H, W) int16 slices -> two (patch_n, P, P) int16 tensors, positions
    from numpy.random.randint as in the reference."""
    assert full_input_img.shape == full_target_img.shape
    h, w = full_input_img.shape
    off = torch.from_numpy(_host_offsets(patch_n, h, w, patch_size)).to(full_input_img.device)
    return _crop(full_input_img.reshape(1, h, w).contiguous(), full_target_img.reshape(1, h, w).contiguous(), 0, 1,
                 patch_n, patch_size, off, 0, 0, False)


class ct_dataset:
    """src/loader.py:7-57 with the slices resident on the device.  load_mode 0 (lazy reads in the reference) and
    1 (all in memory) both load everything here: the point of this class is that pixel data never returns to the
    host.  transform is tested for truth as the reference does (`if self.transform:`, src/loader.py:46): None and
    False - the default of train.py's --transform flag (src/train.py:41,64) - mean no transform; a true value would
    be a host callable on numpy arrays, which the device-resident pipeline does not run."""

    def __init__(self, mode, load_mode, saved_path, test_patient, patch_n=None, patch_size=None, transform=None,
                 device="cuda"):
        assert mode in ['train', 'test'], "mode is 'train' or 'test'"
        assert load_mode in [0, 1], "load_mode is 0 or 1"
        if transform:
            raise NotImplementedError("ct_dataset on the device takes no host transform")
        input_path = sorted(glob(os.path.join(saved_path, '*_input.npy')))
        target_path = sorted(glob(os.path.join(saved_path, '*_target.npy')))
        keep = (lambda f: test_patient not in f) if mode == 'train' else (lambda f: test_patient in f)
        input_ = [f for f in input_path if keep(f)]
        target_ = [f for f in target_path if keep(f)]
        assert len(input_) == len(target_), "every input slice needs its target"
        self.load_mode, self.patch_n, self.patch_size, self.transform = load_mode, patch_n, patch_size, transform
        self.files = (input_, target_)
        if len(input_):
            a = np.stack([np.load(f) for f in input_])
            b = np.stack([np.load(f) for f in target_])
            assert a.dtype == np.int16 and b.dtype == np.int16, "slices are int16 HU (src/prep.py:59)"
            assert a.shape == b.shape
        else:
            a = b = np.zeros((0, 1, 1), dtype=np.int16)
        self.input_ = torch.from_numpy(a).to(device)
        self.target_ = torch.from_numpy(b).to(device)

    def __len__(self):
        return self.target_.shape[0]

    def __getitem__(self, idx):
        if idx < 0 or idx >= len(self):
            raise IndexError(idx)
        if self.patch_size:
            return get_patch(self.input_[idx], self.target_[idx], self.patch_n, self.patch_size)
        return self.input_[idx], self.target_[idx]


class DeviceLoader:
    """What get_loader returns: iterates the dataset in order (the reference's DataLoader has shuffle=False,
    src/loader.py:81), batch_size slices per batch, the last batch possibly smaller."""

    def __init__(self, dataset: ct_dataset, batch_size: int, rng: str = "numpy", seed: int = 0):
        assert rng in ("numpy", "device")
        self.dataset, self.batch_size, self.rng, self.seed = dataset, batch_size, rng, seed
        self.draws = 0            # patches drawn so far by the device generator

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _batches(self, out_f32: bool) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        ds = self.dataset
        S, P, pn = len(ds), ds.patch_size, ds.patch_n
        for first in range(0, S, self.batch_size):
            nsel = min(self.batch_size, S - first)
            if not P:             # whole slices (test mode)
                x, y = ds.input_[first:first + nsel], ds.target_[first:first + nsel]
                yield (x.float().unsqueeze(1), y.float().unsqueeze(1)) if out_f32 else (x, y)
                continue
            _, H, W = ds.input_.shape
            off = None
            if self.rng == "numpy":
                off = torch.from_numpy(_host_offsets(nsel * pn, H, W, P)).to(ds.input_.device)
            x, y = _crop(ds.input_, ds.target_, first, nsel, pn, P, off, self.seed, self.draws, out_f32)
            self.draws += nsel * pn
            if out_f32:
                yield x.view(-1, 1, P, P), y.view(-1, 1, P, P)
            else:
                yield x.view(nsel, pn, P, P), y.view(nsel, pn, P, P)

    def __iter__(self):
        """(x, y) as the reference's DataLoader collates them: int16 (batch, patch_n, P, P), or (batch, H, W)
        without patches."""
        return self._batches(False)

    def float_batches(self):
        """(x, y) as src/train.py:136-146 reshapes them: fp32 (batch*patch_n, 1, P, P) or (batch, 1, H, W)."""
        return self._batches(True)


def get_loader(mode='train', load_mode=0, saved_path=None, test_patient='L064', patch_n=None, patch_size=None,
               transform=None, batch_size=32, num_workers=6, device="cuda", rng="numpy", seed=0):
    """src/loader.py:76-82.  num_workers is accepted and ignored: there are no host workers."""
    dataset_ = ct_dataset(mode, load_mode, saved_path, test_patient, patch_n, patch_size, transform, device=device)
    return DeviceLoader(dataset_, batch_size, rng=rng, seed=seed)
