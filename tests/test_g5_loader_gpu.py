"""ried_patch_crop and the device loader against the oracle's restatement of src/loader.py and against patches the
reference itself produced (tests/golden/loader.npz).  Integer data: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import ried_oracle as O
from reid_gan_amd import loader as L
from reid_gan_amd.synth_slices import synth_slices, write_slice_files

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_get_patch_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    inp, tg = synth_slices(1, 96, 80, seed=41)
    np.random.seed(4242)
    px, py = L.get_patch(torch.from_numpy(inp[0]).to(DEV), torch.from_numpy(tg[0]).to(DEV), 5, 32)
    assert px.dtype == torch.int16
    assert np.array_equal(px.cpu().numpy(), g["gp_x"]) and np.array_equal(py.cpu().numpy(), g["gp_y"])


def test_loader_matches_reference(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    write_slice_files(str(tmp_path), ["L019", "L064", "L150"], 3, 96, 96, seed=50)
    np.random.seed(777)
    dl = L.get_loader(mode="train", load_mode=1, saved_path=str(tmp_path), test_patient="L064", patch_n=4,
                      patch_size=32, transform=False, batch_size=4, num_workers=0)    # False: src/train.py:41,64
    assert len(dl.dataset) == 6 and len(dl) == 2
    batches = list(dl)
    assert len(batches) == int(g["train_batches"][0])
    for bi, (x, y) in enumerate(batches):
        assert np.array_equal(x.cpu().numpy(), g[f"train_x{bi}"]) and np.array_equal(y.cpu().numpy(), g[f"train_y{bi}"])
    # the fused fp32 form is the same numbers, reshaped as src/train.py:136-142 does
    np.random.seed(777)
    for bi, (x, y) in enumerate(dl.float_batches()):
        assert x.dtype == torch.float32 and x.shape == (g[f"train_x{bi}"].shape[0] * 4, 1, 32, 32)
        assert np.array_equal(x.cpu().numpy().reshape(g[f"train_x{bi}"].shape), g[f"train_x{bi}"].astype(np.float32))
        assert np.array_equal(y.cpu().numpy().reshape(g[f"train_y{bi}"].shape), g[f"train_y{bi}"].astype(np.float32))
    dt = L.get_loader(mode="test", load_mode=0, saved_path=str(tmp_path), test_patient="L064", batch_size=2)
    tb = list(dt)
    assert len(tb) == int(g["test_batches"][0])
    for bi, (x, y) in enumerate(tb):
        assert np.array_equal(x.cpu().numpy(), g[f"test_x{bi}"]) and np.array_equal(y.cpu().numpy(), g[f"test_y{bi}"])
    item = dl.dataset[1]
    assert item[0].shape == (4, 32, 32)


@pytest.mark.parametrize("H,W,P,nsel,pn", [(512, 512, 64, 8, 16), (130, 70, 64, 3, 5), (96, 96, 92, 2, 3)])
def test_device_generator_matches_oracle(H, W, P, nsel, pn):
    """Positions drawn in the kernel = the oracle's restatement of the generator; patches = numpy crops there."""
    inp, tg = synth_slices(nsel + 2, H, W, seed=61)
    a, b = torch.from_numpy(inp).to(DEV), torch.from_numpy(tg).to(DEV)
    used = torch.full((nsel * pn, 2), -1, dtype=torch.int32, device=DEV)
    x, y = L._crop(a, b, 1, nsel, pn, P, None, 12345, 1000, True, offsets_out=used)
    off = O.device_patch_offsets(12345, 1000, nsel * pn, H, W, P)
    assert np.array_equal(used.cpu().numpy(), off)
    for j in range(nsel * pn):
        s, (t, l) = 1 + j // pn, off[j]
        assert np.array_equal(x[j].cpu().numpy(), inp[s, t:t + P, l:l + P].astype(np.float32))
        assert np.array_equal(y[j].cpu().numpy(), tg[s, t:t + P, l:l + P].astype(np.float32))
    # int16 output, caller-supplied offsets including the corner positions; out-of-range ones are clamped
    o2 = np.array([[0, 0], [H - P, W - P], [H - P - 1, 0], [H, -3]] * ((nsel * pn + 3) // 4), dtype=np.int32)[:nsel * pn]
    xi, yi = L._crop(a, b, 0, nsel, pn, P, torch.from_numpy(o2).to(DEV), 0, 0, False)
    for j in range(nsel * pn):
        t, l = min(max(o2[j, 0], 0), H - P), min(max(o2[j, 1], 0), W - P)
        assert np.array_equal(xi[j].cpu().numpy(), inp[j // pn, t:t + P, l:l + P])
        assert np.array_equal(yi[j].cpu().numpy(), tg[j // pn, t:t + P, l:l + P])


def test_bad_arguments_are_refused():
    inp, tg = synth_slices(2, 64, 64, seed=62)
    a, b = torch.from_numpy(inp).to(DEV), torch.from_numpy(tg).to(DEV)
    with pytest.raises(RuntimeError, match="smaller than the slice"):
        L._crop(a, b, 0, 1, 2, 64, None, 0, 0, True)          # numpy.random.randint(0, 0) raises in the reference
    with pytest.raises(RuntimeError, match="outside"):
        L._crop(a, b, 1, 2, 2, 32, None, 0, 0, True)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        L._crop(a, b, 0, 1, 2, 30, None, 0, 0, True)
