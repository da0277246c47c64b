"""ried_hu_convert / prep.get_pixels_hu on the GPU against the oracle's restatement of src/prep.py:46-59: int16
results must be bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ried_oracle as O
from reid_gan_amd import dicom, loader, prep
from tests.dicom_writer import write_slice


def _slices(arrs, slopes, icpts):
    out = []
    for a, s, i in zip(arrs, slopes, icpts):
        d = dicom.Slice()
        d.pixel_array, d.RescaleSlope, d.RescaleIntercept = a, s, i
        out.append(d)
    return out


@pytest.mark.parametrize("signed", [False, True])
def test_hu_convert_bit_exact(signed):
    r = np.random.default_rng(11 + signed)
    n, h, w = 7, 40, 24
    dt = np.int16 if signed else np.uint16
    lo, hi = (-32768, 32768) if signed else (0, 65536)
    arrs = [r.integers(lo, hi, (h, w)).astype(dt) for _ in range(n)]
    for a in arrs:                                          # the padding value and its neighbours
        a.reshape(-1)[:6] = np.array([-2000, -1999, -2001, 0, 32767, -32768]).astype(np.int16).view(dt) \
            if signed else np.array([63536, 63537, 63535, 0, 32767, 32768], dtype=np.uint16)
    slopes = [1.0, 1.0, 2.0, 0.5, 1.5, 0.25, 1.0]
    icpts = [-1024.0, 0.0, -1024.0, 10.0, -3.0, 7.0, 32767.0]
    got = prep.get_pixels_hu(_slices(arrs, slopes, icpts))
    want = O.get_pixels_hu(arrs, slopes, icpts)
    assert got.dtype == np.int16 and np.array_equal(got, want)


def test_hu_convert_full_series_and_loader(tmp_path):
    """A 208 x 512 x 512 series (the largest one shipped) stays in HBM and feeds the patch loader."""
    r = np.random.default_rng(5)
    n = 208
    stack = r.integers(0, 4096, (n, 512, 512)).astype(np.uint16)
    stack[:, :4, :4] = 63536
    hu = prep.get_pixels_hu_device(_slices(list(stack), [1.0] * n, [-1024.0] * n))
    assert hu.is_cuda and hu.dtype == torch.int16
    want = O.get_pixels_hu(list(stack[:3]) + [stack[-1]], [1.0] * 4, [-1024.0] * 4)
    assert np.array_equal(hu[[0, 1, 2, n - 1]].cpu().numpy(), want)
    assert int(hu.to(torch.int64).sum().item()) == int(stack.astype(np.int64).sum() - 1024 * stack.size
                                                       - 63536 * 16 * n)
    x, y = loader._crop(hu, hu, 0, 2, 4, 64, None, 3, 0, True)
    assert x.shape == (8, 64, 64) and torch.equal(x, y) and x.min().item() >= -1024.0


def test_series_from_files(tmp_path):
    r = np.random.default_rng(9)
    zs = [30.0, 10.0, 20.0, 0.0]
    arrs = [r.integers(0, 3000, (16, 16)).astype(np.uint16) for _ in zs]
    for i, (z, a) in enumerate(zip(zs, arrs)):
        write_slice(str(tmp_path / f"1-{i}.dcm"), a, z, slope=1.0 if i % 2 else 2.0, intercept=-1024.0,
                    explicit=bool(i % 2))
    got = prep.get_pixels_hu(prep.load_scan(str(tmp_path)))
    order = np.argsort(zs)
    want = O.get_pixels_hu([arrs[i] for i in order], [1.0 if i % 2 else 2.0 for i in order], [-1024.0] * 4)
    assert np.array_equal(got, want)


def test_bad_arguments_are_refused():
    with pytest.raises(ValueError):
        prep.get_pixels_hu_device([])
    d = _slices([np.zeros((3, 3), np.uint16)], [1.0], [0.0])
    with pytest.raises(ValueError):
        prep.get_pixels_hu_device(d)                        # 9 pixels: not a multiple of 8
