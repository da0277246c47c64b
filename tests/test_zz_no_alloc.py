"""ried_colsum_static (the entry point that allocates nothing: its partial rows live in a device array built into the
library) against ried_colsum with a caller workspace, through the C ABI.

Tolerance: both entry points run the same two kernels on the same grid; the first kernel adds its threads' fp32
partial sums with LDS float atomics, whose order differs from launch to launch, so two launches of EITHER entry point
agree only to fp32 round-off of the accumulation: |a - b| <= 4e-7 * sum_p |x[p][c]| per column (a few ulp of the
running sum).  Both are also held to 1e-6 * sum |x| of the fp64 column sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from reid_gan_amd._lib import lib

DEV = "cuda"


def _both(npix, C, ld, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(npix, ld, generator=g).to(torch.bfloat16).to(DEV)
    s = torch.cuda.current_stream().cuda_stream
    ref = torch.full((C,), 0.5, dtype=torch.float32, device=DEV)            # the sums are ADDED onto out
    got = ref.clone()
    scratch = torch.empty(lib.ried_colsum_scratch_floats(C), dtype=torch.float32, device=DEV)
    lib.ried_colsum(a.data_ptr(), ld, npix, C, ref.data_ptr(), scratch.data_ptr(), s)
    lib.ried_colsum_static(a.data_ptr(), ld, npix, C, got.data_ptr(), s)
    torch.cuda.synchronize()
    exact = 0.5 + a[:, :C].double().sum(0)
    mag = a[:, :C].double().abs().sum(0)
    return got.cpu().double(), ref.cpu().double(), exact.cpu(), mag.cpu()


def test_static_scratch_entry_point_twice_second_call_smaller():
    # two calls in a row on the one device array; the second is smaller in both pixels and channels, so it reads
    # fewer rows and columns than the first left behind
    for npix, C, ld, seed in ((128 * 32 * 32, 512, 512, 1), (4 * 8 * 8, 64, 96, 2)):
        got, ref, exact, mag = _both(npix, C, ld, seed)
        assert bool(((got - ref).abs() <= 4e-7 * mag + 1e-7).all()), (npix, C, ((got - ref).abs() / mag).max().item())
        assert bool(((got - exact).abs() <= 1e-6 * mag + 1e-6).all()), (npix, C, ((got - exact).abs() / mag).max().item())
        assert bool(((ref - exact).abs() <= 1e-6 * mag + 1e-6).all()), (npix, C, ((ref - exact).abs() / mag).max().item())


def test_static_scratch_entry_point_refuses_what_does_not_fit():
    a = torch.zeros(8, 2048, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(2048, dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError, match="exceeds the built-in scratch"):
        lib.ried_colsum_static(a.data_ptr(), 2048, 8, 2048, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
