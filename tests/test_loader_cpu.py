"""The oracle's restatement of src/loader.py against patches the reference itself produced
(tests/golden/loader.npz, made by tests/golden/make_golden_loader.py).  Integer data: bit-exact."""
import os

import numpy as np
import pytest

from oracle import ried_oracle as O
from reid_gan_amd.synth_slices import synth_slices


def test_get_patch_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    inp, tg = synth_slices(1, 96, 80, seed=41)
    np.random.seed(4242)
    px, py = O.get_patch(inp[0], tg[0], 5, 32)
    assert px.dtype == np.int16 and np.array_equal(px, g["gp_x"]) and np.array_equal(py, g["gp_y"])


def test_loader_batches_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    # the files the golden script wrote: patients L019, L064, L150 (seeds 50, 51, 52), L064 held out for training
    a0, b0 = synth_slices(3, 96, 96, seed=50)
    a2, b2 = synth_slices(3, 96, 96, seed=52)
    inputs, targets = list(a0) + list(a2), list(b0) + list(b2)      # sorted file order: L019_0.., L150_0..
    np.random.seed(777)
    batches = list(O.loader_batches(inputs, targets, 4, 32, 4))
    assert len(batches) == int(g["train_batches"][0]) == 2
    for bi, (x, y) in enumerate(batches):
        assert np.array_equal(x, g[f"train_x{bi}"]) and np.array_equal(y, g[f"train_y{bi}"])
    a1, b1 = synth_slices(3, 96, 96, seed=51)                       # test mode: whole slices of the held-out patient
    assert np.array_equal(a1[:2], g["test_x0"]) and np.array_equal(b1[2:], g["test_y1"])


def test_device_offsets_are_in_range_and_spread():
    off = O.device_patch_offsets(seed=9, first_draw=0, n=4096, h=512, w=300, patch_size=64)
    assert off[:, 0].min() >= 0 and off[:, 0].max() < 512 - 64 and off[:, 1].min() >= 0 and off[:, 1].max() < 300 - 64
    assert off[:, 0].max() > 430 and off[:, 1].max() > 225 and len(np.unique(off, axis=0)) > 4000
    again = O.device_patch_offsets(seed=9, first_draw=100, n=10, h=512, w=300, patch_size=64)
    assert np.array_equal(again, off[100:110])                       # a pure function of (seed, draw)


def test_transform_flag_is_tested_for_truth_like_the_reference(tmp_path):
    """train.py passes transform=args.transform, whose default is False (src/train.py:41,64), and the reference's
    dataset tests `if self.transform:` (src/loader.py:46): False and None both mean "no transform"."""
    from reid_gan_amd import loader as L
    from reid_gan_amd.synth_slices import write_slice_files
    write_slice_files(str(tmp_path), ["L019", "L064"], 2, 96, 96, seed=50)
    for tr in (False, None):
        ds = L.ct_dataset("train", 1, str(tmp_path), "L064", patch_n=4, patch_size=32, transform=tr, device="cpu")
        assert len(ds) == 2 and ds.input_.dtype.is_floating_point is False
    with pytest.raises(NotImplementedError):
        L.ct_dataset("train", 1, str(tmp_path), "L064", transform=lambda a: a, device="cpu")
