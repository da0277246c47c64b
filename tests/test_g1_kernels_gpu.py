"""Kernel-level parity on the GPU: each C-ABI entry point against a torch CPU fp32 statement of the same op.

Inputs are rounded to bf16 first, so the only differences are the fp32 accumulation order and the bf16
rounding of the output; the tolerance (1.5e-2 of the output's max, 3e-3 for fp32 outputs) is stated per check.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from reid_gan_amd import ops
from reid_gan_amd._lib import lib
from reid_gan_amd.ops import View

DEV = "cuda"


def r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()


def to_view(x_nchw) -> View:
    t = x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    n, h, w, c = t.shape
    return View(t, t.data_ptr(), c, c, n, h, w)


def from_view(v: View):
    return v.torch().float().cpu().permute(0, 3, 1, 2).contiguous()


def close(a, b, tol=1.5e-2, what=""):
    a, b = a.double(), b.double()
    scale = max(b.abs().max().item(), 1e-20)
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.4g} vs scale {scale:.4g}"


def pack(w_oikk):
    """(Cout,Cin,kh,kw) fp32 -> bf16 [Cout][kh][kw][Cin] through the library's pack kernel."""
    co, ci, kh, kw = w_oikk.shape
    src = w_oikk.contiguous().to(DEV)
    dst = torch.empty(co * kh * kw * ci, dtype=torch.bfloat16, device=DEV)
    ops.pack_weight(src, dst, co, kh, kw, ci, ci * kh * kw, kw, 1, kh * kw)
    return dst


CONV_CASES = [
    # cin, cout, k, stride, pad, H
    (64, 128, (3, 3), 1, (1, 1), 16),
    (32, 64, (3, 3), 2, (1, 1), 16),
    (16, 32, (3, 3), 1, (1, 1), 8),
    (48, 48, (3, 3), 1, (1, 1), 8),
    (256, 64, (1, 1), 1, (0, 0), 8),
    (64, 64, (1, 7), 1, (0, 3), 8),
    (128, 128, (7, 1), 1, (3, 0), 4),
    (224, 192, (3, 3), 1, (1, 1), 4),
    (512, 160, (1, 1), 1, (0, 0), 4),
    (64, 16, (1, 1), 1, (0, 0), 16),
]


@pytest.mark.parametrize("cin,cout,k,s,p,H", CONV_CASES)
def test_conv2d_forward_stats(cin, cout, k, s, p, H):
    x = r(3, cin, H, H, seed=1).abs()
    w = r(cout, cin, *k, seed=2, scale=(cin * k[0] * k[1]) ** -0.5)
    ref = F.conv2d(x, w, None, s, p)
    xv = to_view(x)
    Ho, Wo = ref.shape[-2:]
    yv = ops.new_view(3, Ho, Wo, cout)
    rows = ops.conv2d_stat_rows(3, Ho, Wo, cin, cout, k[0], k[1], s, s)
    stats = torch.full((rows, 2, cout), float("nan"), device=DEV)        # every row must be overwritten
    ops.conv2d(xv, pack(w), yv, k[0], k[1], s, s, p[0], p[1], stats=stats)
    close(from_view(yv), ref, what="y")
    st = stats.sum(0)
    close(st[0].cpu(), ref.sum((0, 2, 3)), 3e-3 * 10, "sum")
    close(st[1].cpu(), (ref * ref).sum((0, 2, 3)), 3e-3, "sumsq")
    save = torch.empty(2, cout, device=DEV)
    ops.bn_finalize(stats, rows, cout, ref.numel() // cout, save)
    close(save[0].cpu(), ref.mean((0, 2, 3)), 3e-3 * 10, "mean")
    close(save[1].cpu(), torch.rsqrt(ref.var((0, 2, 3), unbiased=False) + 1e-5), 3e-3, "invstd")
    # the fused form: the conv's last block finalizes; twice, to see the ticket counter return to zero
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    rm, rv = torch.zeros(cout, device=DEV), torch.ones(cout, device=DEV)
    nbt = torch.zeros(1, dtype=torch.int64, device=DEV)
    for it in range(2):
        save2 = torch.full((2, cout), float("nan"), device=DEV)
        y2 = ops.new_view(3, Ho, Wo, cout)
        ops.conv2d_bn(xv, pack(w), y2, k[0], k[1], s, s, p[0], p[1], stats, counter, save2, rm, rv, nbt)
        torch.cuda.synchronize()
        assert counter.item() == 0 and nbt.item() == it + 1
        close(from_view(y2), ref, what="y (fused)")
        close(save2[0].cpu(), save[0].cpu(), 1e-5, "mean (fused)")
        close(save2[1].cpu(), save[1].cpu(), 1e-5, "invstd (fused)")
    n = ref.numel() // cout
    m, v = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False) * n / (n - 1)
    close(rm.cpu(), 0.19 * m, 3e-2, "running mean after two steps")
    close(rv.cpu(), 0.81 + 0.19 * v, 3e-3, "running var after two steps")


HALO_CASES = [
    # cin, cout, N, H, W, transposed
    (64, 64, 2, 16, 32, False),
    (128, 192, 1, 32, 16, False),       # a second, partial tile of output channels
    (256, 72, 3, 16, 16, False),
    (64, 128, 2, 32, 32, True),         # the data gradient of a 3x3 s1 p1 conv
    (192, 64, 1, 16, 48, True),
    (512, 128, 4, 8, 8, False),         # 8x8 tiles of four images per block
    (128, 72, 8, 8, 24, True),
    (64, 64, 12, 24, 8, False),
]


@pytest.mark.parametrize("cin,cout,N,H,W,tr", HALO_CASES)
def test_conv2d_halo_tile_kernel(cin, cout, N, H, W, tr):
    """3x3 s1 p1 layers on maps of whole 16x16 tiles take the halo-tile kernel: against torch, and every epilogue
    feature against the general kernel (ried_conv2d_force_tile(-1) turns the halo kernel off)."""
    x = r(N, cin, H, W, seed=50)
    b = r(cout, seed=52)
    if tr:
        w = r(cin, cout, 3, 3, seed=51, scale=(cin * 9) ** -0.5)
        ref = F.conv_transpose2d(x, w, None, 1, 1)
        wp = pack(w.permute(1, 0, 2, 3))
    else:
        w = r(cout, cin, 3, 3, seed=51, scale=(cin * 9) ** -0.5)
        ref = F.conv2d(x, w, None, 1, 1)
        wp = pack(w)
    xv = to_view(x)
    rows = ops.conv2d_stat_rows(N, H, W, cin, cout, 3, 3, 1, 1, tr)
    out = {}
    for name, force in (("halo", 0), ("general", -1)):
        lib.ried_conv2d_force_tile(force, 0, 0)
        try:
            yv = ops.new_view(N, H, W, cout)
            stats = torch.full((rows, 2, cout), float("nan"), device=DEV)     # every row must be defined afterwards
            ops.conv2d(xv, wp, yv, 3, 3, 1, 1, 1, 1, transposed=tr, stats=stats)
            # bias + ReLU + fp32 copy into a channel slice, then accumulate + mask + post-mask column sums
            big = ops.new_view(N, H, W, cout + 16, zero=True)
            y2 = torch.zeros(N, H, W, cout, device=DEV)
            ops.conv2d(xv, wp, big.slice(8, 8 + cout), 3, 3, 1, 1, 1, 1, transposed=tr, bias=b.to(DEV), act=1, y2=y2)
            acc = to_view(ref * 0.5)
            mask = to_view(r(N, cout, H, W, seed=53))
            st2 = torch.full((rows, 2, cout), float("nan"), device=DEV)
            ops.conv2d(xv, wp, acc, 3, 3, 1, 1, 1, 1, transposed=tr, accumulate=True, mask=mask, mask_slope=0.2,
                       stats=st2, stats_post=True)
            fold = torch.zeros(ops.bn_fold_rows(), 2, cout, device=DEV)
            ops.conv2d(xv, wp, ops.new_view(N, H, W, cout), 3, 3, 1, 1, 1, 1, transposed=tr, stats=fold, stats_post=2)
            out[name] = (from_view(yv), stats.sum(0).cpu(), from_view(big), y2.cpu(), from_view(acc), st2.sum(0).cpu(),
                         fold.sum(0).cpu())
        finally:
            lib.ried_conv2d_force_tile(0, 0, 0)
    y, st, bg, y2, ac, s2, fo = out["halo"]
    close(y, ref, what="y")
    close(st[0], ref.sum((0, 2, 3)), 3e-2, "sum")
    close(st[1], (ref * ref).sum((0, 2, 3)), 3e-3, "sumsq")
    want = F.relu(ref + b.view(1, -1, 1, 1))
    close(bg[:, 8:8 + cout], want, what="bias + relu into a slice")
    assert bg[:, :8].abs().max() == 0 and bg[:, 8 + cout:].abs().max() == 0
    close(y2.permute(0, 3, 1, 2), want, 1e-3, "fp32 copy")
    m = r(N, cout, H, W, seed=53).to(torch.bfloat16).float()
    wanted = (ref + (ref * 0.5).to(torch.bfloat16).float()) * torch.where(m > 0, 1.0, 0.2)
    close(ac, wanted, what="accumulate + mask")
    close(s2[0], wanted.sum((0, 2, 3)), 3e-2, "post-mask column sums")
    close(fo[1], (ref * ref).sum((0, 2, 3)), 3e-3, "folded sumsq")
    for a, g, what in zip(out["halo"], out["general"], ("y", "stats", "slice", "y2", "acc", "stats2", "fold")):
        close(a, g, 4e-3, what + " vs general kernel")


@pytest.mark.parametrize("cin,cout,N,H,W", [(128, 64, 2, 16, 16), (256, 128, 1, 32, 16), (64, 72, 1, 16, 32),
                                             (512, 64, 3, 16, 16)])
def test_conv2d_stride2_data_gradient_halo_kernel(cin, cout, N, H, W):
    """The data gradient of a 3x3 s2 p1 conv (H x W -> 2H x 2W) on whole tiles takes the four-phase halo kernel:
    against torch and, with every epilogue feature the discriminator uses, against the general kernel."""
    dy = r(N, cin, H, W, seed=60)
    w = r(cin, cout, 3, 3, seed=61, scale=(cin * 9) ** -0.5)               # conv weight (Cout_conv=cin, Cin_conv=cout)
    ref = F.conv_transpose2d(dy, w, None, 2, 1, output_padding=1)
    wp = pack(w.permute(1, 0, 2, 3))
    dyv = to_view(dy)
    rows = ops.conv2d_stat_rows(N, 2 * H, 2 * W, cin, cout, 3, 3, 2, 2, True)
    xfwd = r(N, cout, 2 * H, 2 * W, seed=62)                                # forward output of the layer below
    out = {}
    for name, force in (("halo", 0), ("general", -1)):
        lib.ried_conv2d_force_tile(force, 0, 0)
        try:
            yv = ops.new_view(N, 2 * H, 2 * W, cout)
            stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
            ops.conv2d(dyv, wp, yv, 3, 3, 2, 2, 1, 1, transposed=True, stats=stats)
            gv = ops.new_view(N, 2 * H, 2 * W, cout)
            st2 = torch.full((rows, 2, cout), float("nan"), device=DEV)
            ops.conv2d(dyv, wp, gv, 3, 3, 2, 2, 1, 1, transposed=True, mask=to_view(xfwd), mask_slope=0.2, stats=st2,
                       stats_post=True)
            out[name] = (from_view(yv), stats.sum(0).cpu(), from_view(gv), st2.sum(0).cpu())
        finally:
            lib.ried_conv2d_force_tile(0, 0, 0)
    y, st, g, s2 = out["halo"]
    close(y, ref, what="y")
    close(st[1], (ref * ref).sum((0, 2, 3)), 3e-3, "sumsq over phases")
    xm = xfwd.to(torch.bfloat16).float()
    want = ref * torch.where(xm > 0, 1.0, 0.2)
    close(g, want, what="masked gradient")
    close(s2[0], want.sum((0, 2, 3)), 3e-2, "bias-gradient column sums")
    for a, b, what in zip(out["halo"], out["general"], ("y", "stats", "masked", "column sums")):
        close(a, b, 4e-3, what + " vs general kernel")


@pytest.mark.parametrize("th", [16, 8])
@pytest.mark.parametrize("cin,cout,N,H,W", [(64, 128, 2, 64, 64), (128, 256, 1, 32, 64), (32, 64, 2, 32, 32),
                                             (96, 192, 1, 32, 32), (64, 72, 3, 32, 32)])
def test_conv2d_stride2_forward_halo_kernel(cin, cout, N, H, W, th):
    """3x3 stride-2 pad-1 forward on maps of whole 16x16 OUTPUT tiles takes the half-slab halo kernel (Cin a multiple
    of 32; 64 or 128 output channels per block, rows past Cout are zeros): against torch with bias + LeakyReLU (the
    discriminator's layers), and with BatchNorm statistics (plain rows, folded rows, a channel slice of a wider
    buffer) against the general kernel."""
    x = r(N, cin, H, W, seed=80)
    w = r(cout, cin, 3, 3, seed=81, scale=(cin * 9) ** -0.5)
    b = r(cout, seed=82, scale=0.1)
    raw = F.conv2d(x, w, None, 2, 1)
    ref = F.leaky_relu(raw + b.view(1, -1, 1, 1), 0.2)
    wp = pack(w)
    xv = to_view(x)
    Ho, Wo = H // 2, W // 2
    rows = ops.conv2d_stat_rows(N, Ho, Wo, cin, cout, 3, 3, 2, 2, False)
    out = {}
    # (0, 1, 0) / (0, 2, 0): the halo kernel with 16 x 16 / 8 x 16 output tiles (it is off by default)
    for name, force in (("halo", (0, 1 if th == 16 else 2, 0)), ("general", (-1, 0, 0))):
        lib.ried_conv2d_force_tile(*force)
        try:
            assert ops.conv2d_which(xv, wp, ops.new_view(N, Ho, Wo, cout), 3, 3, 2, 2, 1, 1).startswith(
                "conv_halo_s2f_kernel" if name == "halo" else "conv_glds_kernel")
            yv = ops.new_view(N, Ho, Wo, cout)
            ops.conv2d(xv, wp, yv, 3, 3, 2, 2, 1, 1, bias=b.to(DEV), act=2)
            rv = ops.new_view(N, Ho, Wo, cout)
            stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
            ops.conv2d(xv, wp, rv, 3, 3, 2, 2, 1, 1, stats=stats)
            big = ops.new_view(N, Ho, Wo, cout + 16, zero=True)
            fold = torch.zeros(ops.bn_fold_rows(), 2, cout, device=DEV)
            ops.conv2d(xv, wp, big.slice(8, 8 + cout), 3, 3, 2, 2, 1, 1, stats=fold, stats_post=2)
            out[name] = (from_view(yv), from_view(rv), stats.sum(0).cpu(), from_view(big), fold.sum(0).cpu())
        finally:
            lib.ried_conv2d_force_tile(0, 0, 0)
    y, yr, st, yb, fo = out["halo"]
    close(y, ref, what="bias + LeakyReLU")
    close(yr, raw, what="raw")
    close(st[0], raw.sum((0, 2, 3)), 3e-3, "sum")
    close(st[1], (raw * raw).sum((0, 2, 3)), 3e-3, "sum of squares")
    close(yb[:, 8:8 + cout], raw, what="slice")
    assert yb[:, :8].abs().max() == 0 and yb[:, 8 + cout:].abs().max() == 0
    close(fo[1], (raw * raw).sum((0, 2, 3)), 3e-3, "folded sum of squares")
    for a, c, what in zip(out["halo"], out["general"], ("y", "raw", "stats", "slice", "folded stats")):
        close(a, c, 4e-3, what + " vs general kernel")


@pytest.mark.parametrize("cin,cout,N,H,W", [(192, 64, 2, 16, 16), (96, 32, 1, 32, 16), (64, 48, 3, 16, 32),
                                             (32, 128, 1, 16, 16), (384, 128, 1, 16, 16), (96, 24, 2, 16, 16)])
def test_deconv_4x4_stride2_halo_kernel(cin, cout, N, H, W):
    """ConvTranspose2d(4, 2, 1) on maps of whole 16x16 tiles takes the four-phase halo kernel (Cin a multiple of
    32: the half-slab tail is skipped): against torch, and with BatchNorm statistics (plain rows and folded)
    against the general kernel."""
    x = r(N, cin, H, W, seed=70)
    w = r(cin, cout, 4, 4, seed=71, scale=(cin * 4) ** -0.5)                # ConvTranspose2d layout (Cin,Cout,kh,kw)
    ref = F.conv_transpose2d(x, w, None, 2, 1)
    wp = pack(w.permute(1, 0, 2, 3))
    xv = to_view(x)
    rows = ops.conv2d_stat_rows(N, 2 * H, 2 * W, cin, cout, 4, 4, 2, 2, True)
    out = {}
    for name, force in (("halo", (0, 1, 0)), ("general", (-1, 0, 0))):   # (0, 1, 0): the kernel whatever the block count
        lib.ried_conv2d_force_tile(*force)
        try:
            yv = ops.new_view(N, 2 * H, 2 * W, cout)
            stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
            ops.conv2d(xv, wp, yv, 4, 4, 2, 2, 1, 1, transposed=True, stats=stats)
            big = ops.new_view(N, 2 * H, 2 * W, cout + 16, zero=True)       # a channel slice of a wider buffer
            fold = torch.zeros(ops.bn_fold_rows(), 2, cout, device=DEV)
            ops.conv2d(xv, wp, big.slice(8, 8 + cout), 4, 4, 2, 2, 1, 1, transposed=True, stats=fold, stats_post=2)
            out[name] = (from_view(yv), stats.sum(0).cpu(), from_view(big), fold.sum(0).cpu())
        finally:
            lib.ried_conv2d_force_tile(0, 0, 0)
    y, st, yb, fo = out["halo"]
    close(y, ref, what="y")
    close(st[0], ref.sum((0, 2, 3)), 3e-3, "sum over phases")
    close(st[1], (ref * ref).sum((0, 2, 3)), 3e-3, "sum of squares over phases")
    close(yb[:, 8:8 + cout], ref, what="slice")
    assert yb[:, :8].abs().max() == 0 and yb[:, 8 + cout:].abs().max() == 0
    close(fo[1], (ref * ref).sum((0, 2, 3)), 3e-3, "folded sum of squares")
    for a, b, what in zip(out["halo"], out["general"], ("y", "stats", "slice", "folded stats")):
        close(a, b, 4e-3, what + " vs general kernel")


@pytest.mark.parametrize("cin,cout,k,s,tr,H", [(64, 64, 3, 1, False, 32), (64, 32, 1, 1, False, 16), (48, 40, 3, 2, False, 32),
                                               (96, 32, 4, 2, True, 16), (128, 64, 3, 1, False, 16)])
def test_per_image_statistics_rows_and_apply(cin, cout, k, s, tr, H):
    """stats_post = 3: one (sum, sum of squares) row per image from every convolution kernel (general, halo-tile,
    four-phase), then ried_bn_finalize_images / ried_bn_apply_images = train-mode BatchNorm + ReLU of each image on
    its own (the batch-1 evaluation loop of src/test.py:107-131 on a batch)."""
    N, p = 3, (1 if k > 1 else 0)
    x = r(N, cin, H, H, seed=80)
    x[1] = x[1] * 3 + 1                                     # images with different statistics
    if tr:
        w = r(cin, cout, k, k, seed=81, scale=(cin * k) ** -0.5)
        ref = F.conv_transpose2d(x, w, None, s, p)
        wp = pack(w.permute(1, 0, 2, 3))
    else:
        w = r(cout, cin, k, k, seed=81, scale=(cin * k * k) ** -0.5)
        ref = F.conv2d(x, w, None, s, p)
        wp = pack(w)
    Ho = ref.shape[-1]
    for force in ((0, 1, 0), (-1, 0, 0)):                   # with and without the halo-tile kernels
        lib.ried_conv2d_force_tile(*force)
        try:
            yv = ops.new_view(N, Ho, Ho, cout)
            stats = torch.zeros(N, 2, cout, device=DEV)
            ops.conv2d(to_view(x), wp, yv, k, k, s, s, p, p, transposed=tr, stats=stats, stats_post=3)
        finally:
            lib.ried_conv2d_force_tile(0, 0, 0)
        close(stats[:, 0].cpu(), ref.sum((2, 3)), 3e-3, "per-image sums")
        close(stats[:, 1].cpu(), (ref * ref).sum((2, 3)), 3e-3, "per-image sums of squares")
    save = torch.empty(N, 2, cout, device=DEV)
    ops.bn_finalize_images(stats, 1, N, cout, Ho * Ho, save)
    gamma, beta = 1 + 0.1 * r(cout, seed=82), 0.1 * r(cout, seed=83)
    out = ops.new_view(N, Ho, Ho, cout)
    ops.bn_apply_images(yv, out, save, gamma.to(DEV), beta.to(DEV), relu=True)
    raw = from_view(yv)
    want = torch.cat([F.relu(F.batch_norm(raw[i:i + 1], None, None, gamma, beta, True, 0.1, 1e-5)) for i in range(N)])
    close(from_view(out), want, 2e-2, "per-image BatchNorm + ReLU")


def test_bn_finalize_folds_many_rows():
    """Thousands of partial rows (512x512 inference) go through the two-step reduction: same statistics."""
    g = torch.Generator().manual_seed(3)
    rows, C = 5000, 40
    stats = torch.rand(rows, 2, C, generator=g).to(DEV)
    stats[:, 1] += 1.0                                            # keeps the variance positive
    a, b = torch.empty(2, C, device=DEV), torch.empty(2, C, device=DEV)
    ops.bn_finalize(stats, rows, C, rows, a)
    fold = torch.full((ops.bn_finalize_fold_floats(C),), float("nan"), device=DEV)
    ops.bn_finalize(stats, rows, C, rows, b, fold=fold)
    close(b.cpu(), a.cpu(), 1e-5, "folded vs direct")
    close(a[0].cpu(), stats[:, 0].sum(0).cpu() / rows, 1e-5, "mean")


@pytest.mark.parametrize("rows", [37, 5000])
def test_bn_finalize_on_a_channel_slice_of_wider_rows(rows):
    """A layer that is a channel slice of a wider convolution (ldstat): same statistics and running statistics as on
    a compact copy of its columns, directly and through the two-step reduction, bit for bit."""
    g = torch.Generator().manual_seed(4)
    total, c0, C = 208, 128, 48
    wide = torch.rand(rows, 2, total, generator=g).to(DEV)
    wide[:, 1] += 1.0
    compact = wide[:, :, c0:c0 + C].contiguous()
    for use_fold in (False, True):
        res = []
        for stats, ld in ((compact, 0), (wide.view(-1)[c0:], total)):
            save = torch.full((2, C), float("nan"), device=DEV)
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            nbt = torch.zeros((), dtype=torch.long, device=DEV)
            fold = torch.full((ops.bn_finalize_fold_floats(C),), float("nan"), device=DEV) if use_fold else None
            ops.bn_finalize(stats, rows, C, rows, save, rm, rv, nbt, fold=fold, ldstat=ld)
            res.append((save, rm, rv, nbt))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        assert not torch.isnan(res[1][0]).any() and res[1][3].item() == 1


def test_conv2d_bias_act_slice_accumulate():
    x = r(2, 64, 8, 8, seed=3)
    w = r(128, 64, 3, 3, seed=4, scale=0.05)
    b = r(128, seed=5)
    ref = F.leaky_relu(F.conv2d(x, w, b, 2, 1), 0.2)
    big = ops.new_view(2, 4, 4, 256, zero=True)
    ops.conv2d(to_view(x), pack(w), big.slice(64, 192), 3, 3, 2, 2, 1, 1, bias=b.to(DEV), act=2)
    got = from_view(big)
    close(got[:, 64:192], ref, what="slice")
    assert got[:, :64].abs().max() == 0 and got[:, 192:].abs().max() == 0
    ops.conv2d(to_view(x), pack(w), big.slice(64, 192), 3, 3, 2, 2, 1, 1, bias=b.to(DEV), act=2, accumulate=True)
    close(from_view(big)[:, 64:192], 2 * ref, what="accumulate")


TCONV_CASES = [
    # cin, cout, k, stride, pad, H(in), out_pad
    (96, 32, (4, 4), 2, (1, 1), 8, 0),
    (192, 64, (4, 4), 2, (1, 1), 4, 0),
    (64, 32, (3, 3), 2, (1, 1), 8, 1),      # data gradient of a 3x3 s2 p1 conv (16 -> 8)
    (128, 128, (1, 7), 1, (0, 3), 4, 0),
    (48, 32, (3, 3), 1, (1, 1), 8, 0),
    (1024, 64, (1, 1), 1, (0, 0), 4, 0),
]


@pytest.mark.parametrize("cin,cout,k,s,p,H,op", TCONV_CASES)
def test_conv2d_transposed(cin, cout, k, s, p, H, op):
    x = r(2, cin, H, H, seed=6)
    w = r(cin, cout, *k, seed=7, scale=(cin * k[0] * k[1]) ** -0.5)       # ConvTranspose2d layout
    ref = F.conv_transpose2d(x, w, None, s, p, output_padding=op)
    Ho, Wo = ref.shape[-2:]
    yv = ops.new_view(2, Ho, Wo, cout)
    wp = pack(w.permute(1, 0, 2, 3))                                      # -> [Cout][kh][kw][Cin]
    rows = ops.conv2d_stat_rows(2, Ho, Wo, cin, cout, k[0], k[1], s, s, True)
    stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
    ops.conv2d(to_view(x), wp, yv, k[0], k[1], s, s, p[0], p[1], transposed=True, stats=stats)
    close(from_view(yv), ref, what="y")
    close(stats.sum(0)[1].cpu(), (ref * ref).sum((0, 2, 3)), 3e-3, "sumsq over phases")


WGRAD_CASES = [
    (64, 128, (3, 3), 1, (1, 1), 8, 5),
    (64, 128, (3, 3), 2, (1, 1), 16, 7),
    (48, 48, (3, 3), 1, (1, 1), 8, 1),
    # 3 x 3, at most 64 input channels, enough output pixels: a kernel row per block
    (24, 40, (3, 3), 1, (1, 1), 224, 5),
    (64, 136, (3, 3), 2, (1, 1), 384, 37),
    (8, 64, (3, 3), 2, (1, 1), 450, 3),
    (32, 64, (3, 3), 2, (1, 1), 16, 3),
    (16, 32, (3, 3), 1, (1, 1), 8, 2),
    (256, 64, (1, 1), 1, (0, 0), 8, 4),
    (64, 64, (1, 7), 1, (0, 3), 8, 1),
    (192, 192, (3, 1), 1, (1, 0), 4, 2),
    (512, 1024, (1, 1), 1, (0, 0), 4, 1),
]


@pytest.mark.parametrize("cin,cout,k,s,p,H,splitk", WGRAD_CASES)
def test_conv2d_wgrad(cin, cout, k, s, p, H, splitk):
    x = r(3, cin, H, H, seed=8).requires_grad_(False)
    w = r(cout, cin, *k, seed=9).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = r(*y.shape, seed=10)
    (ref,) = torch.autograd.grad(y, w, dy)
    taps = k[0] * k[1]
    slab = torch.empty(splitk * taps * cout * cin, device=DEV)
    ops.conv2d_wgrad(to_view(x), to_view(dy), k[0], k[1], s, s, p[0], p[1], slab, splitk)
    out = torch.zeros(cout, cin, *k, device=DEV)
    ops.wgrad_reduce(slab, splitk, taps, cout, cin, out, 1, cin * taps, taps)
    close(out.cpu(), ref, 3e-3, "dW")
    if H >= 200:
        assert lib.ried_conv2d_wgrad_taps_per_block(cin, cout, k[0], k[1], y.shape[0] * y.shape[2] * y.shape[3]) == 3


@pytest.mark.parametrize("cin,cout,H,N,nt", [(96, 32, 8, 2, 1), (96, 32, 128, 3, 2), (40, 64, 160, 2, 2)])
def test_deconv_wgrad(cin, cout, H, N, nt):
    """nt: taps of the 4 x 4 window per block (two where every block still has enough pixels to walk)."""
    assert lib.ried_conv2d_wgrad_taps_per_block(cout, cin, 4, 4, N * H * H) == nt
    x = r(N, cin, H, H, seed=11)
    w = r(cin, cout, 4, 4, seed=12).requires_grad_(True)
    y = F.conv_transpose2d(x, w, None, 2, 1)
    dy = r(*y.shape, seed=13)
    (ref,) = torch.autograd.grad(y, w, dy)
    slab = torch.empty(4 * 16 * cin * cout, device=DEV)
    # gathered tensor = dy (the deconv output side), pixel-indexed tensor = x
    ops.conv2d_wgrad(to_view(dy), to_view(x), 4, 4, 2, 2, 1, 1, slab, 4)
    out = torch.zeros(cin, cout, 4, 4, device=DEV)
    ops.wgrad_reduce(slab, 4, 16, cin, cout, out, 1, cout * 16, 16)
    close(out.cpu(), ref, 3e-3, "dW")


@pytest.mark.parametrize("cin,cout,s,slope,acc", [(128, 64, 2, 0.2, False), (64, 72, 1, 0.0, True),
                                                  (512, 256, 2, 0.2, True)])
def test_conv2d_fused_activation_backward_and_bias_sums(cin, cout, s, slope, acc):
    """Data gradient of a strided 3x3 conv with the (Leaky)ReLU backward of the tensor it produces and the column
    sums of the masked result (the bias gradient of the layer below) fused into the epilogue."""
    N, H = 3, 8
    g = r(N, cin, H, H, seed=41)                                   # gradient w.r.t. the conv output
    w = r(cin, cout, 3, 3, seed=42, scale=(cin * 9) ** -0.5)       # the conv's weight is (cin_out=cin, cout_in=cout)
    Ho = H * s
    dx = F.conv_transpose2d(g, w, None, s, 1, output_padding=s - 1)
    yact = r(N, cout, Ho, Ho, seed=43)                             # forward output of the layer below
    old = r(N, cout, Ho, Ho, seed=44)
    ref = (dx + (old if acc else 0)) * torch.where(yact > 0, 1.0, slope)
    yv = to_view(old)
    rows = ops.conv2d_stat_rows(N, Ho, Ho, cin, cout, 3, 3, s, s, True)
    stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
    ops.conv2d(to_view(g), pack(w.permute(1, 0, 2, 3)), yv, 3, 3, s, s, 1, 1, transposed=True, accumulate=acc,
               mask=to_view(yact), mask_slope=slope, stats=stats, stats_post=True)
    close(from_view(yv), ref, what="masked dgrad")
    bias = torch.zeros(cout, device=DEV)
    ops.stat_rows_sum(stats, rows, cout, bias)
    close(bias.cpu(), ref.sum((0, 2, 3)), 2e-2, "bias gradient")


@pytest.mark.parametrize("cin,cout,N,H,bf16", [(64, 64, 2, 16, False), (64, 64, 2, 16, True), (128, 256, 3, 8, True),
                                               (256, 512, 4, 4, True), (256, 512, 4, 4, False)])
def test_tap_layer_with_the_perceptual_mse_fused(cin, cout, N, H, bf16):
    """A VGG tap layer on the prediction images: y = relu(conv + bias), d = y (fp32) - target, column sums of d and
    d^2, gout = gscale * d - against torch, with the target as an fp32 tensor or as stored bf16 activations (halo
    kernels on 16 x 16 and 8 x 8 maps, the general kernel on 4 x 4)."""
    x = r(N, cin, H, H, seed=71)
    w = r(cout, cin, 3, 3, seed=72, scale=(cin * 9) ** -0.5)
    b = r(cout, seed=73) * 0.1
    ref = F.relu(F.conv2d(x, w, b, 1, 1))
    tgt = (ref + r(N, cout, H, H, seed=74) * 0.05).clamp_min(0).to(torch.bfloat16).float()      # bf16-exact either way
    d = ref - tgt
    yv, gv = ops.new_view(N, H, H, cout), ops.new_view(N, H, H, cout)
    rows = ops.conv2d_stat_rows(N, H, H, cin, cout, 3, 3)
    stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
    gscale = 0.37
    if bf16:
        ops.conv2d_tap_mse_bf16(to_view(x), pack(w), b.to(DEV), yv, to_view(tgt), gv, gscale, stats)
    else:
        t32 = tgt.permute(0, 2, 3, 1).contiguous().to(DEV)
        ops.conv2d_tap_mse(to_view(x), pack(w), b.to(DEV), yv, t32, gv, gscale, stats)
    torch.cuda.synchronize()
    close(from_view(yv), ref, what="tap output")
    got = from_view(gv)
    assert (got - gscale * d).abs().max() <= 1.5e-2 * ref.abs().max() * gscale + 4e-3 * (gscale * d).abs().max()
    tot = stats.sum(0).cpu()
    assert torch.isfinite(tot).all()
    close(tot[0], d.sum((0, 2, 3)), 2e-2, "sum d")
    close(tot[1], (d * d).sum((0, 2, 3)), 2e-2, "sum d^2")


def _sign_bytes(y_nchw: torch.Tensor) -> torch.Tensor:
    """uint8 [pixels][C/8]: bit q of byte c set where channel 8c + q is positive."""
    N, C, H, W = y_nchw.shape
    bits = (y_nchw.permute(0, 2, 3, 1).reshape(-1, C // 8, 8) > 0).to(torch.int32)
    return (bits << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8)


@pytest.mark.parametrize("cin,cout,s,slope,N,H", [(128, 64, 2, 0.2, 3, 16), (128, 64, 2, 0.2, 2, 8), (64, 72, 1, 0.0, 3, 8),
                                                  (64, 32, 2, 0.2, 1, 16)])
def test_conv2d_activation_backward_from_sign_bytes(cin, cout, s, slope, N, H):
    """The same fused data gradient with the mask given as one bit per value: bit-identical to the bf16 mask (the
    stride-2 halo kernel's staged epilogue at 16 x 16 maps, the general kernel's two epilogues otherwise)."""
    g = r(N, cin, H, H, seed=45)
    w = r(cin, cout, 3, 3, seed=46, scale=(cin * 9) ** -0.5)
    Ho = H * s
    yact = r(N, cout, Ho, Ho, seed=47)
    rows = ops.conv2d_stat_rows(N, Ho, Ho, cin, cout, 3, 3, s, s, True)
    outs = []
    for bits in (False, True):
        yv = ops.new_view(N, Ho, Ho, cout)
        stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
        if bits:
            ops.conv2d_signs(to_view(g), pack(w.permute(1, 0, 2, 3)), yv, 3, 3, s, s, 1, 1,
                             _sign_bytes(yact).to(DEV).contiguous(), transposed=True, mask_slope=slope, stats=stats,
                             stats_post=True)
        else:
            ops.conv2d(to_view(g), pack(w.permute(1, 0, 2, 3)), yv, 3, 3, s, s, 1, 1, transposed=True,
                       mask=to_view(yact), mask_slope=slope, stats=stats, stats_post=True)
        bias = torch.zeros(cout, device=DEV)
        ops.stat_rows_sum(stats, rows, cout, bias)
        outs.append((from_view(yv), bias.cpu()))
    ref = F.conv_transpose2d(g, w, None, s, 1, output_padding=s - 1) * torch.where(yact > 0, 1.0, slope)
    close(outs[1][0], ref, what="masked dgrad from sign bytes")
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("cin,N,H,W,bits", [(128, 2, 16, 16, True), (128, 3, 32, 16, False), (64, 1, 16, 48, True)])
def test_stride2_data_gradient_with_the_first_layer_weight_gradient_folded(cin, N, H, W, bits):
    """ried_conv2d_s2_dgrad_c1fold against the two launches it replaces (masked data gradient stored, then
    ried_conv_c1_wgrad on it) and against torch: the bias sums are identical, the weight gradient differs only by the
    summation order and by the image being carried as two bf16 parts (16 mantissa bits) instead of fp32."""
    slope = 0.2
    g = r(N, cin, H, W, seed=61)
    w = r(cin, 64, 3, 3, seed=62, scale=(cin * 9) ** -0.5)
    wt = pack(w.permute(1, 0, 2, 3))
    yact = r(N, 64, 2 * H, 2 * W, seed=63)
    ximg = (torch.randn(N, 1, 2 * H, 2 * W, generator=torch.Generator().manual_seed(64)) * 300 - 200).to(DEV)
    rows = ops.conv2d_stat_rows(N, 2 * H, 2 * W, cin, 64, 3, 3, 2, 2, True)
    mask = _sign_bytes(yact).to(DEV).contiguous() if bits else to_view(yact)
    # the two launches
    yv = ops.new_view(N, 2 * H, 2 * W, 64)
    stats_a = torch.full((rows, 2, 64), float("nan"), device=DEV)
    if bits:
        ops.conv2d_signs(to_view(g), wt, yv, 3, 3, 2, 2, 1, 1, mask, transposed=True, mask_slope=slope, stats=stats_a,
                         stats_post=True)
    else:
        ops.conv2d(to_view(g), wt, yv, 3, 3, 2, 2, 1, 1, transposed=True, mask=mask, mask_slope=slope, stats=stats_a,
                   stats_post=True)
    dw_a = torch.zeros(64, 9, device=DEV)
    scr = torch.full((ops.conv_c1_wgrad_scratch_floats(64),), float("nan"), device=DEV)
    ops.conv_c1_wgrad(ximg, yv, 1, dw_a, scr, in_scale=1 / 512.0, in_shift=0.3)
    # the fold
    stats_b = torch.full((rows, 2, 64), float("nan"), device=DEV)
    dw_b = torch.full((64, 9), 0.5, device=DEV)
    scr2 = torch.full((ops.conv2d_s2_dgrad_c1fold_scratch_floats(N, H, W),), float("nan"), device=DEV)
    ops.conv2d_s2_dgrad_c1fold(to_view(g), wt, mask, slope, stats_b, ximg, dw_b, scr2, in_scale=1 / 512.0, in_shift=0.3)
    ba, bb = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    ops.stat_rows_sum(stats_a, rows, 64, ba)
    ops.stat_rows_sum(stats_b, rows, 64, bb)
    assert torch.equal(ba, bb)
    close(dw_b.cpu() - 0.5, dw_a.cpu(), 2e-4, "folded weight gradient vs two launches")
    # torch: the masked gradient rounded to bf16 as both paths do, then the first layer's weight gradient in fp32
    da = F.conv_transpose2d(g, w, None, 2, 1, output_padding=1) * torch.where(yact > 0, 1.0, slope)
    da = da.to(torch.bfloat16).float()
    xa = (ximg.cpu() / 512.0 + 0.3).requires_grad_(False)
    w0 = torch.zeros(64, 1, 3, 3, requires_grad=True)
    (gw,) = torch.autograd.grad(F.conv2d(xa, w0, None, 1, 1), w0, da)
    close(dw_b.cpu().reshape(64, 1, 3, 3) - 0.5, gw, 2e-3, "folded weight gradient vs torch")


def test_maxpool_bwd_overwrite_and_relu_mask():
    x = r(2, 16, 8, 8, seed=51)
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(torch.relu(xr), 2, 2)
    gy = r(*y.shape, seed=52)
    (gx,) = torch.autograd.grad(y, xr, gy)                       # ReLU backward included
    xpost = torch.relu(x)
    dxv = to_view(torch.full_like(x, 7.0))                        # garbage that must be overwritten
    ops.pool_bwd(to_view(xpost), to_view(gy), dxv, ops.POOL_MAX2S2, accumulate=False, relu_mask=True)
    # where several window elements tie at 0 after the ReLU the mask zeroes them anyway
    close(from_view(dxv), gx, what="maxpool+relu backward")


def test_job_table_pack_and_reduce_match_single_calls():
    """One launch over a job table = the per-layer calls."""
    g = torch.Generator().manual_seed(5)
    jobs_p, jobs_r = ops.JobTable(13), ops.JobTable(12)
    keep, checks = [], []
    sigma = torch.tensor([1.7], device=DEV)
    for n, (co, ci, kh, kw) in enumerate([(64, 32, 3, 3), (96, 192, 4, 4), (8, 8, 1, 1), (520, 24, 1, 7)]):
        t = kh * kw
        w = torch.randn(co, ci, kh, kw, generator=g).to(DEV)
        ref = torch.empty(co * t * ci, dtype=torch.bfloat16, device=DEV)
        got = torch.full_like(ref, float("nan"))
        sg = sigma if n % 2 else None
        ops.pack_weight(w, ref, co, kh, kw, ci, ci * t, kw, 1, t, sg)
        jobs_p.add([w.data_ptr(), got.data_ptr(), 0 if sg is None else sg.data_ptr(), co, kh, kw, ci, ci * t, kw, 1, t,
                    kh * kw * ci, 0], co * t * ci)
        splitk = 3 + n
        slab = torch.randn(splitk, t, co, ci, generator=g).to(DEV)
        oref = torch.zeros(co, ci, kh, kw, device=DEV)
        ogot = torch.full_like(oref, 0.5)
        ops.wgrad_reduce(slab, splitk, t, co, ci, oref, 1, ci * t, t)
        jobs_r.add([slab.data_ptr(), ogot.data_ptr(), splitk, t * co * ci, t, co, ci, 1, ci * t, t, 0, 0], nblocks=co)
        keep += [w, slab]
        checks += [(ref, got), (oref, ogot)]
    # a master laid out [tap][p][g] (not the [p][g][tap] rows of conv weights): the scattered-store path
    slab = torch.randn(5, 4, 24, 40, generator=g).to(DEV)
    oref, ogot = torch.zeros(4, 24, 40, device=DEV), torch.full((4, 24, 40), 0.25, device=DEV)
    ops.wgrad_reduce(slab, 5, 4, 24, 40, oref, 24 * 40, 40, 1)
    jobs_r.add([slab.data_ptr(), ogot.data_ptr(), 5, 4 * 24 * 40, 4, 24, 40, 24 * 40, 40, 1, 0, 0], nblocks=24)
    keep.append(slab); checks.append((oref, ogot))
    # an fp32 destination with a row pitch (a [rows][pitch] matrix filled in column blocks)
    src = torch.randn(40, 24, generator=g).to(DEV)
    dst = torch.zeros(40, 64, device=DEV)
    jobs_p.add([src.data_ptr(), dst.data_ptr() + 4 * 8, 0, 40, 1, 1, 24, 24, 1, 1, 1, 64, 1], 40 * 24)
    jobs_p.finalize(DEV); jobs_r.finalize(DEV)
    ops.pack_jobs(jobs_p)
    ops.wgrad_reduce_jobs(jobs_r)
    torch.cuda.synchronize()
    assert torch.equal(checks[0][0], checks[0][1])               # packing: the same arithmetic per element
    for a, b in checks:                                           # reductions: the splits are summed in another order
        close(a.float().cpu(), b.float().cpu(), 1e-6, "job table vs single call")
    assert torch.equal(dst[:, 8:32], src) and dst[:, :8].abs().sum() == 0 and dst[:, 32:].abs().sum() == 0



@pytest.mark.parametrize("a_tr,b_tr", [(0, 0), (1, 0), (1, 1), (0, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (64, 64, 64), (32, 256, 256), (512, 64, 64), (256, 1024, 32)])
def test_bgemm(a_tr, b_tr, M, N, K):
    batch = 3
    A = r(batch, M, K, seed=14)
    B = r(batch, N, K, seed=15)
    ref = torch.einsum("bmk,bnk->bnm", A, B)
    Ad = (A.transpose(1, 2) if a_tr else A).contiguous().to(torch.bfloat16).to(DEV)
    Bd = (B.transpose(1, 2) if b_tr else B).contiguous().to(torch.bfloat16).to(DEV)
    out = torch.empty(batch, N, M, device=DEV)
    ops.bgemm(Ad.data_ptr(), M if a_tr else K, M * K, a_tr, Bd.data_ptr(), N if b_tr else K, N * K, b_tr, M, N, K,
              batch, out.data_ptr(), M, N * M, out_f32=True)
    close(out.cpu(), ref, 3e-3, "C")


def test_bgemm_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed accumulator write."""
    M = N = K = 64
    A = torch.eye(M).unsqueeze(0)
    B = (torch.arange(N * K).reshape(1, N, K) % 251).float()
    out = torch.empty(1, N, M, device=DEV)
    Ad, Bd = A.to(torch.bfloat16).to(DEV), B.to(torch.bfloat16).to(DEV)     # keep the storages alive
    ops.bgemm(Ad.data_ptr(), K, M * K, 0, Bd.data_ptr(), K, N * K, 0, M, N, K, 1, out.data_ptr(), M, N * M,
              out_f32=True)
    assert torch.equal(out.cpu(), torch.einsum("bmk,bnk->bnm", A, B))


def test_bn_forward_backward():
    N, C, H = 4, 48, 8
    x = (r(N, C, H, H, seed=16) * 2 + 0.5).to(torch.bfloat16).float()
    gamma, beta = r(C, seed=17) + 1.0, r(C, seed=18) * 0.3
    xr = x.clone().requires_grad_(True)
    g2, b2 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.relu(F.batch_norm(xr, rm, rv, g2, b2, True, 0.1, 1e-5))
    dy = r(*y.shape, seed=19)
    y.backward(dy)
    xv = to_view(x)
    stats = torch.stack([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))]).to(DEV)
    save = torch.empty(2, C, device=DEV)
    rm_d, rv_d = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    ops.bn_finalize(stats, 1, C, N * H * H, save, rm_d, rv_d, nbt)
    yv = ops.new_view(N, H, H, C)
    ops.bn_apply(xv, yv, save, gamma.to(DEV), beta.to(DEV))
    close(from_view(yv), y.detach(), what="y")
    close(rm_d.cpu(), rm, 1e-4, "running_mean")
    close(rv_d.cpu(), rv, 1e-4, "running_var")
    assert nbt.item() == 1
    dxv = ops.new_view(N, H, H, C)
    scratch = torch.full((ops.bn_backward_scratch_floats(C),), float("nan"), device=DEV)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_backward(to_view(dy), xv, dxv, save, gamma.to(DEV), beta.to(DEV), scratch, dg, db)
    close(from_view(dxv), xr.grad, what="dx")
    close(dg.cpu(), g2.grad, 5e-3, "dgamma")
    close(db.cpu(), b2.grad, 5e-3, "dbeta")
    # with a ticket counter the reduction's last block finishes the sums: same results, counter back at zero
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    for _ in range(2):
        dx2 = ops.new_view(N, H, H, C)
        dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        scratch.fill_(float("nan"))
        ops.bn_backward(to_view(dy), xv, dx2, save, gamma.to(DEV), beta.to(DEV), scratch, dg2, db2, counter=counter)
        torch.cuda.synchronize()
        assert counter.item() == 0
        assert torch.equal(dx2.torch(), dxv.torch()) and torch.equal(dg2, dg) and torch.equal(db2, db)
    # the folded form: sums added into a few zeroed rows with float atomics, no finishing launch, no scratch;
    # parameter gradients accumulate when asked to
    folded = torch.zeros(ops.bn_backward_folded_floats(C), device=DEV)
    dx3 = ops.new_view(N, H, H, C)
    dg3, db3 = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    ops.bn_backward(to_view(dy), xv, dx3, save, gamma.to(DEV), beta.to(DEV), None, dg3, db3, accumulate=True,
                    folded=folded)
    close(from_view(dx3), xr.grad, what="dx (folded)")
    close(dg3.cpu() - 1, g2.grad, 5e-3, "dgamma (folded)")
    close(db3.cpu() - 1, b2.grad, 5e-3, "dbeta (folded)")
    assert (dx3.torch().float() - dxv.torch().float()).abs().max() <= 2e-2 * dxv.torch().float().abs().max()


def _bn_case(N, C, H, seed):
    x = (r(N, C, H, H, seed=seed) * 2 + 0.5).to(torch.bfloat16).float()
    gamma, beta = r(C, seed=seed + 1) + 1.0, r(C, seed=seed + 2) * 0.3
    dy = r(N, C, H, H, seed=seed + 3).to(torch.bfloat16).float()
    mean, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
    save = torch.stack([mean, (var + 1e-5).rsqrt()]).to(DEV).contiguous()
    return x, dy, gamma.to(DEV), beta.to(DEV), save


@pytest.mark.parametrize("N,C,H", [(4, 48, 8), (2, 8, 4), (32, 64, 16), (128, 64, 32), (16, 1024, 4)])
def test_bn_backward_in_one_launch(N, C, H):
    """folded + counter: partial sums, grid barrier and apply in one launch (the register-resident variant for at
    most four passes per thread, the re-reading one above), against the two-launch folded form; repeated with the
    same (re-zeroed) rows, and refused for a layer that is too large."""
    assert ops.bn_backward_fused_ok(N * H * H, C)
    x, dy, gamma, beta, save = _bn_case(N, C, H, 400 + C)
    xv, dyv = to_view(x), to_view(dy)
    nf = ops.bn_backward_folded_floats(C)
    ref_dx = ops.new_view(N, H, H, C)
    rg, rb = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_backward(dyv, xv, ref_dx, save, gamma, beta, None, rg, rb, folded=torch.zeros(nf, device=DEV))
    buf = torch.zeros(nf + 8, device=DEV)
    for rep in range(2):
        buf.zero_()
        dx = ops.new_view(N, H, H, C)
        dx.torch().fill_(float("nan"))
        dg, db = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
        ops.bn_backward(dyv, xv, dx, save, gamma, beta, None, dg, db, accumulate=True, folded=buf, counter=buf[nf:])
        torch.cuda.synchronize()
        a, b = dx.torch().float(), ref_dx.torch().float()
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 2e-2 * b.abs().max() + 1e-6, (rep, (a - b).abs().max().item())
        close(dg.cpu() - 1, rg.cpu(), 5e-3, "dgamma")
        close(db.cpu() - 1, rb.cpu(), 5e-3, "dbeta")
    assert ops.barrier_timeouts() == 0
    assert not ops.bn_backward_fused_ok(128 * 64 * 64, 64)
    with pytest.raises(RuntimeError, match="too large"):
        big = ops.new_view(128, 64, 64, 64)
        ops.bn_backward(big, big, big, save, gamma, beta, None, folded=buf, counter=buf[nf:])


def test_bn_backward_jobs_in_one_launch():
    """Job table with all_folded = 2: layers of different widths sharing the pixel count, one launch."""
    N, H = 8, 8
    Cs = [64, 96, 16, 384]
    assert ops.bn_backward_fused_ok(N * H * H, max(Cs), len(Cs))
    rows, keep, want = [], [], []
    for i, C in enumerate(Cs):
        x, dy, gamma, beta, save = _bn_case(N, C, H, 500 + 10 * i)
        xv, dyv = to_view(x), to_view(dy)
        nf = ops.bn_backward_folded_floats(C)
        ref_dx = ops.new_view(N, H, H, C)
        rg, rb = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        ops.bn_backward(dyv, xv, ref_dx, save, gamma, beta, None, rg, rb, folded=torch.zeros(nf, device=DEV))
        buf = torch.zeros(nf + 8, device=DEV)
        dx = ops.new_view(N, H, H, C)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        rows.append(ops.bn_backward_job(dyv, xv, dx, save, gamma, beta, buf[nf:], dg, db, act=1, folded=buf))
        keep.append((xv, dyv, gamma, beta, save, buf))
        want.append((dx, ref_dx, dg, rg, db, rb))
    table = torch.tensor(rows, dtype=torch.int64).to(DEV)
    ops.bn_backward_jobs(table, len(Cs), N * H * H, max(Cs), 2)
    torch.cuda.synchronize()
    for dx, ref_dx, dg, rg, db, rb in want:
        a, b = dx.torch().float(), ref_dx.torch().float()
        assert (a - b).abs().max() <= 2e-2 * b.abs().max() + 1e-6
        close(dg.cpu(), rg.cpu(), 5e-3, "dgamma")
        close(db.cpu(), rb.cpu(), 5e-3, "dbeta")
    assert ops.barrier_timeouts() == 0


@pytest.mark.parametrize("C,cin,H", [(48, 32, 8), (160, 64, 8), (16, 64, 16), (1024, 64, 4)])
def test_bn_small_layer_kernels(C, cin, H):
    """conv with folded (atomic) statistics -> bn_apply_train, against torch and the finalize + apply pair."""
    N = 8
    x = (r(N, cin, H, H, seed=30) * 0.7).to(torch.bfloat16).float()
    w = (r(C, cin, 1, 1, seed=31) * 0.2).to(torch.bfloat16).float()
    gamma, beta = r(C, seed=32) + 1.0, r(C, seed=33) * 0.3
    wp = w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    raw = ops.new_view(N, H, H, C)
    nrows = min(ops.bn_fold_rows(), ops.conv2d_stat_rows(N, H, H, cin, C, 1, 1))
    stats = torch.zeros(nrows * 2 * C, device=DEV)
    ops.conv2d(to_view(x), wp, raw, 1, 1, 1, 1, 0, 0, stats=stats, stats_post=2)
    rawf = from_view(raw)                                   # the bf16-rounded conv result, NCHW fp32
    ref_raw = F.conv2d(x, w)
    close(rawf, ref_raw, 1e-2, "conv")
    tot = stats.view(nrows, 2, C).sum(0).cpu()
    close(tot[0], ref_raw.sum((0, 2, 3)), 1e-2, "folded sum")
    close(tot[1], (ref_raw * ref_raw).sum((0, 2, 3)), 1e-2, "folded sum of squares")
    # BatchNorm of the stored (rounded) conv output, statistics from the unrounded accumulators
    xr = rawf.clone().requires_grad_(True)
    g2, b2 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.relu(F.batch_norm(xr, rm, rv, g2, b2, True, 0.1, 1e-5))
    save = torch.empty(2, C, device=DEV)
    rm_d, rv_d = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    nbt = torch.zeros((), dtype=torch.long, device=DEV)
    yv = ops.new_view(N, H, H, C)
    ops.bn_apply_train(raw, yv, stats, nrows, save, gamma.to(DEV), beta.to(DEV), rm_d, rv_d, nbt)
    close(from_view(yv), y.detach(), 2e-2, "y")
    n = N * H * H                    # the running statistics come from the unrounded accumulators
    close(rm_d.cpu(), 0.1 * ref_raw.mean((0, 2, 3)), 2e-3, "running_mean")
    close(rv_d.cpu(), 0.9 + 0.1 * ref_raw.var((0, 2, 3), unbiased=True), 2e-3, "running_var")
    assert nbt.item() == 1
    # same as the three-launch kernels on the same statistics
    save3 = torch.empty(2, C, device=DEV)
    ops.bn_finalize(stats, nrows, C, N * H * H, save3)
    close(save.cpu(), save3.cpu(), 1e-5, "save")
    y3 = ops.new_view(N, H, H, C)
    ops.bn_apply(raw, y3, save3, gamma.to(DEV), beta.to(DEV))
    close(from_view(yv), from_view(y3), 1e-2, "y vs three-launch")


def test_bn_apply_train_job_table_matches_the_per_layer_launches():
    """Sibling layers as channel slices of one wide 1x1 convolution: the job-table BatchNorm forward gives, bit for
    bit, what one bn_apply_train per layer gives (outputs, save, running statistics, batch counter)."""
    N, H, cin, widths, acts = 8, 8, 64, (128, 32, 16, 48), (0, 1, 1, 2)
    total = sum(widths)
    x = (r(N, cin, H, H, seed=40) * 0.7).to(torch.bfloat16).float()
    wp = (r(total, 1, 1, cin, seed=41) * 0.2).to(torch.bfloat16).to(DEV)
    raw = ops.new_view(N, H, H, total)
    nrows = min(ops.bn_fold_rows(), ops.conv2d_stat_rows(N, H, H, cin, total, 1, 1))
    stats = torch.zeros(ops.bn_fold_rows() * 2 * total, device=DEV)
    ops.conv2d(to_view(x), wp, raw, 1, 1, 1, 1, 0, 0, stats=stats, stats_post=2)

    def state(C, seed):
        return dict(save=torch.full((2 * C,), float("nan"), device=DEV), rm=(r(C, seed=seed) * 0.1).to(DEV),
                    rv=(r(C, seed=seed + 1).abs() + 0.5).to(DEV), nbt=torch.full((), 3, dtype=torch.long, device=DEV),
                    gamma=(r(C, seed=seed + 2) + 1.0).to(DEV), beta=(r(C, seed=seed + 3) * 0.3).to(DEV))

    outs, rows, c0 = [], [], 0
    for i, (C, act) in enumerate(zip(widths, acts)):
        a, b = state(C, 50 + 10 * i), state(C, 50 + 10 * i)
        ya, yb = ops.new_view(N, H, H, C), ops.new_view(N, H, H, C)
        xs = raw.slice(c0, c0 + C)
        ops.bn_apply_train(xs, ya, stats[c0:], nrows, a["save"], a["gamma"], a["beta"], a["rm"], a["rv"], a["nbt"],
                           relu=act, ldstat=total)
        rows.append(ops.bn_apply_train_job(xs, yb, stats.data_ptr() + 4 * c0, nrows, total, b["save"], b["rm"], b["rv"],
                                           b["nbt"], b["gamma"], b["beta"], act=act))
        outs.append((a, b, ya, yb))
        c0 += C
    table = torch.tensor(rows, dtype=torch.int64).to(DEV)
    ops.bn_apply_train_jobs(table, len(rows), N * H * H, max(widths), min(widths))
    torch.cuda.synchronize()
    for a, b, ya, yb in outs:
        assert torch.equal(ya.torch(), yb.torch())
        for k in ("save", "rm", "rv", "nbt"):
            assert torch.equal(a[k], b[k]), k
        assert b["nbt"].item() == 4 and not torch.isnan(b["save"]).any()


@pytest.mark.parametrize("n,d,C", [(256, 32, 256), (64, 64, 512)])
def test_attention_scores_softmax(n, d, C):
    """softmax_j(q_i . k_j) in one pass (src/rinet.py:171-174) against torch on the bf16-rounded q, k."""
    N, side = 3, int(n ** 0.5)
    qkv = r(N, 2 * d + C, side, side, seed=70) * 0.7
    v = to_view(qkv)
    qb = qkv.to(torch.bfloat16).float().permute(0, 2, 3, 1).reshape(N, n, -1)
    ref = torch.softmax(qb[:, :, :d] @ qb[:, :, d:2 * d].transpose(1, 2), dim=-1)
    P = torch.full((N, n, n), float("nan"), dtype=torch.bfloat16, device=DEV)
    assert ops.attn_scores_softmax_supported(n, d) and not ops.attn_scores_softmax_supported(n, d + 8)
    ops.attn_scores_softmax(v, d, P)
    close(P.float().cpu(), ref, 6e-3, "P")
    assert abs(P.float().sum(-1).cpu() - 1).max() < 2e-2


@pytest.mark.parametrize("C,npix", [(64, 1000), (512, 37), (8, 300)])
def test_l2norm_tap(C, npix):
    a = r(npix, C, seed=40).requires_grad_(True)
    b = r(npix, C, seed=41)
    with torch.no_grad():
        a[5] = b[5]                                  # a zero difference: norm 0, gradient 0 (not NaN)
    ref = torch.linalg.norm(a - b, dim=1).sum() * 0.25
    ref.backward()
    ad, bd = a.detach().to(DEV), b.to(DEV)
    loss_d = torch.zeros(1, device=DEV)
    g = torch.full((npix, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    ops.l2norm_tap(ad.data_ptr(), bd.data_ptr(), npix, C, 0.25, loss_d, g.data_ptr(), 0.25)
    close(loss_d.cpu(), ref.detach().reshape(1), 1e-5, "loss")
    close(g.float().cpu(), a.grad, 5e-3, "gradient")
    assert g[5].abs().max().item() == 0.0


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pool(mode):
    N, C, H = 2, 16, 8
    x = r(N, C, H, H, seed=20).requires_grad_(True)
    if mode == 0:
        y = F.max_pool2d(x, 3, 2, 1)
    elif mode == 1:
        y = F.max_pool2d(x, 2, 2)
    else:
        y = F.avg_pool2d(x, 3, 1, 1)
    dy = r(*y.shape, seed=21)
    y.backward(dy)
    xv = to_view(x.detach())
    yv = ops.new_view(N, y.shape[2], y.shape[3], C)
    ops.pool_fwd(xv, yv, mode)
    close(from_view(yv), y.detach(), 4e-3, "y")
    dxv = ops.new_view(N, H, H, C, zero=True)
    ops.pool_bwd(None if mode == 2 else xv, to_view(dy), dxv, mode)
    close(from_view(dxv), x.grad, 1e-2, "dx")


@pytest.mark.parametrize("N,C,H,W", [(4, 64, 128, 64), (2, 256, 64, 128)])
def test_average_pool_on_large_maps(N, C, H, W):
    """Maps of a million 8-channel chunks or more take the strip kernel (inference sizes): same result, borders
    included, in a channel slice of a wider buffer."""
    x = r(N, C, H, W, seed=22)
    y = F.avg_pool2d(x, 3, 1, 1)
    big = ops.new_view(N, H, W, C + 16, zero=True)
    ops.pool_fwd(to_view(x), big.slice(8, 8 + C), 2)
    got = from_view(big)
    close(got[:, 8:8 + C], y, 4e-3, "strip average pool")
    assert got[:, :8].abs().max() == 0 and got[:, 8 + C:].abs().max() == 0


@pytest.mark.parametrize("N,C,H,W", [(2, 16, 16, 24), (3, 96, 8, 8), (2, 64, 64, 32)])
def test_average_pool_with_batchnorm_statistics(N, C, H, W):
    """ried_avgpool3_stats (the pool of a branch_pool applied AFTER its 1x1 convolution): y = avg_pool2d(x, 3, 1, 1) as
    F.avg_pool2d computes it, and the partial rows reduce to the column sums of y and y^2 (fp32, before rounding):
    1e-3 relative.  Through ried_bn_finalize the rows give the mean and 1 / sqrt(var + eps) of the pooled tensor.
    And the commutation the engine relies on: pool(conv1x1(x)) == conv1x1(pool(x)) for a bias-free convolution."""
    x = r(N, C, H, W, seed=23)
    y = F.avg_pool2d(x, 3, 1, 1)
    xv, yv = to_view(x), ops.new_view(N, H, W, C)
    rows = ops.avgpool3_stats_rows(N * H * W, C)
    stats = torch.full((rows, 2, C), float("nan"), device=DEV)
    ops.avgpool3_stats(xv, yv, stats)
    close(from_view(yv), y, 4e-3, "y")
    s1, s2 = stats[:, 0].sum(0).cpu(), stats[:, 1].sum(0).cpu()
    npix = N * H * W
    assert torch.allclose(s1 / npix, y.mean((0, 2, 3)), rtol=1e-3, atol=1e-4)
    assert torch.allclose(s2 / npix, (y * y).mean((0, 2, 3)), rtol=1e-3, atol=1e-5)
    save = torch.empty(2 * C, device=DEV)
    ops.bn_finalize(stats, rows, C, npix, save)
    var = y.var((0, 2, 3), unbiased=False)
    assert torch.allclose(save[:C].cpu(), y.mean((0, 2, 3)), rtol=1e-3, atol=1e-4)
    assert torch.allclose(save[C:].cpu(), 1.0 / torch.sqrt(var + 1e-5), rtol=2e-3)
    w = r(8, C, 1, 1, seed=24)
    assert torch.allclose(F.avg_pool2d(F.conv2d(x, w), 3, 1, 1), F.conv2d(F.avg_pool2d(x, 3, 1, 1), w), rtol=1e-4, atol=1e-5)


def test_maxpool_ties_first_max():
    x = torch.zeros(1, 8, 4, 4)
    y = F.max_pool2d(x.clone().requires_grad_(True), 3, 2, 1)
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr, 3, 2, 1).backward(torch.ones_like(y))
    dxv = ops.new_view(1, 4, 4, 8, zero=True)
    ops.pool_bwd(to_view(x), to_view(torch.ones_like(y)), dxv, 0)
    assert torch.equal(from_view(dxv), xr.grad)


def test_upsample():
    N, C, H = 2, 24, 4
    x = r(N, C, H, H, seed=22).requires_grad_(True)
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    dy = r(*y.shape, seed=23)
    y.backward(dy)
    yv = ops.new_view(N, 2 * H, 2 * H, C)
    ops.upsample2x_fwd(to_view(x.detach()), yv)
    close(from_view(yv), y.detach(), 5e-3, "y")
    dxv = ops.new_view(N, H, H, C, zero=True)
    ops.upsample2x_bwd(to_view(dy), dxv)
    close(from_view(dxv), x.grad, 1e-2, "dx")


def test_softmax_fwd_bwd():
    rows, n = 37, 256
    S = torch.randn(rows, n, generator=torch.Generator().manual_seed(1)) * 3
    Sr = S.clone().requires_grad_(True)
    P = torch.softmax(Sr, -1)
    dP = torch.randn(rows, n, generator=torch.Generator().manual_seed(2))
    P.backward(dP)
    Pd = torch.empty(rows, n, dtype=torch.bfloat16, device=DEV)
    ops.softmax_fwd(S.to(DEV), Pd, rows, n)
    close(Pd.float().cpu(), P.detach(), 5e-3, "P")
    dS = torch.empty(rows, n, dtype=torch.bfloat16, device=DEV)
    ops.softmax_bwd(dP.to(DEV), Pd, dS, rows, n)
    close(dS.float().cpu(), Sr.grad, 2e-2, "dS")


def test_adam_matches_torch():
    n = 1003
    p0 = torch.randn(n, generator=torch.Generator().manual_seed(3))
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=4e-4, betas=(0.01, 0.99))
    pd = torch.zeros(1004, device=DEV)[:n]
    pd.copy_(p0)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in range(1, 4):
        g = torch.randn(n, generator=torch.Generator().manual_seed(10 + step))
        p.grad = g.clone()
        opt.step()
        ops.adam(pd, g.to(DEV), m, v, 4e-4, cnt)
        assert torch.allclose(pd.cpu(), p.detach(), rtol=2e-5, atol=1e-7)
    assert cnt.item() == 3


@pytest.mark.parametrize("cout,stride", [(32, 2), (64, 1)])
def test_conv_c1(cout, stride):
    N, H = 3, 16
    x = torch.randn(N, 1, H, H, generator=torch.Generator().manual_seed(4)) * 500 - 400
    w = r(cout, 1, 3, 3, seed=24).requires_grad_(True)
    b = r(cout, seed=25)
    sigma = torch.tensor([1.7])
    xa = x / 4096.0 + 0.25
    y = F.leaky_relu(F.conv2d(xa, w / sigma, b, stride, 1), 0.2)
    Ho = y.shape[-1]
    yv = ops.new_view(N, Ho, Ho, cout)
    rows = ops.conv_c1_stat_rows(N, H, H, cout, stride)
    stats = torch.full((rows, 2, cout), float("nan"), device=DEV)
    ops.conv_c1_fwd(x.to(DEV), w.detach().to(DEV), yv, stride, sigma=sigma.to(DEV), bias=b.to(DEV),
                    in_scale=1 / 4096.0, in_shift=0.25, act=2, stats=stats)
    close(from_view(yv), y.detach(), what="y")
    # the same call also writing one sign bit per stored value
    yv2 = ops.new_view(N, Ho, Ho, cout)
    signs = torch.full((N * Ho * Ho, cout // 8), 0xA5, dtype=torch.uint8, device=DEV)
    ops.conv_c1_fwd(x.to(DEV), w.detach().to(DEV), yv2, stride, sigma=sigma.to(DEV), bias=b.to(DEV),
                    in_scale=1 / 4096.0, in_shift=0.25, act=2, signs=signs)
    assert torch.equal(from_view(yv2), from_view(yv))
    assert torch.equal(signs.cpu(), _sign_bytes(from_view(yv)))
    raw = F.conv2d(xa, w.detach() / sigma, None, stride, 1)
    close(stats.sum(0)[0].cpu(), raw.sum((0, 2, 3)), 1e-3, "sum")
    dy = r(*y.shape, seed=26)
    raw2 = F.conv2d(xa, w, None, stride, 1)
    (gw,) = torch.autograd.grad(raw2, w, dy)
    dw = torch.zeros(cout, 9, device=DEV)
    scr = torch.full((ops.conv_c1_wgrad_scratch_floats(cout),), float("nan"), device=DEV)
    ops.conv_c1_wgrad(x.to(DEV), to_view(dy), stride, dw, scr, in_scale=1 / 4096.0, in_shift=0.25)
    close(dw.cpu().reshape(cout, 1, 3, 3), gw, 3e-3, "dw")
    if stride == 1:
        xr = xa.clone().requires_grad_(True)
        (gx,) = torch.autograd.grad(F.conv2d(xr, w.detach() / sigma, None, 1, 1), xr, dy)
        dx = torch.ones(N, 1, H, H, device=DEV)
        ops.conv_c1_dgrad(to_view(dy), w.detach().to(DEV), dx, sigma=sigma.to(DEV), scale=0.5, accumulate=True)
        close(dx.cpu() - 1.0, 0.5 * gx, 3e-3, "dx")


@pytest.mark.parametrize("H,W,acc", [(16, 32, True), (8, 64, False), (5, 12, True), (7, 9, False)])
def test_conv_c1_dgrad_strips(H, W, acc):
    """Four-pixel strips when the width allows, single pixels otherwise: against torch."""
    N = 3
    w = r(64, 1, 3, 3, seed=80)
    sigma = torch.tensor([1.3])
    dy = r(N, 64, H, W, seed=81)
    xr = torch.zeros(N, 1, H, W, requires_grad=True)
    (gx,) = torch.autograd.grad(F.conv2d(xr, w / sigma, None, 1, 1), xr, dy.to(torch.bfloat16).float())
    dx = torch.full((N, 1, H, W), 2.0 if acc else float("nan"), device=DEV)
    ops.conv_c1_dgrad(to_view(dy), w.to(DEV), dx, sigma=sigma.to(DEV), scale=0.25, accumulate=acc)
    close(dx.cpu() - (2.0 if acc else 0.0), 0.25 * gx, 2e-3, "dx")


def test_out_conv():
    N, H = 2, 16
    x = torch.randn(N, 1, H, H, generator=torch.Generator().manual_seed(5)) * 500
    h = r(N, 32, H, H, seed=27).abs()
    w = r(1, 33, 1, 1, seed=28).requires_grad_(True)
    hr = h.clone().requires_grad_(True)
    out = F.conv2d(torch.cat([x, hr], 1), w)
    dout = torch.randn(N, 1, H, H, generator=torch.Generator().manual_seed(6))
    out.backward(dout)
    od = torch.empty(N, 1, H, H, device=DEV)
    hv = to_view(h)
    ops.out_conv_fwd(x.to(DEV), hv, w.detach().reshape(33).to(DEV), od)
    close(od.cpu(), out.detach(), 1e-5, "out")
    # with BatchNorm + ReLU of the raw tensor applied on the fly (the forward-only generator plan)
    raw = r(N, 32, H, H, seed=29)
    save = torch.cat([raw.mean((0, 2, 3)), torch.rsqrt(raw.var((0, 2, 3), unbiased=False) + 1e-5)])
    gamma, beta = 1 + 0.1 * r(32, seed=30), 0.1 * r(32, seed=31)
    hn = F.relu(F.batch_norm(raw, None, None, gamma, beta, True, 0.1, 1e-5))
    ops.out_conv_fwd(x.to(DEV), to_view(raw), w.detach().reshape(33).to(DEV), od,
                     bn=(save.to(DEV), gamma.to(DEV), beta.to(DEV)))
    close(od.cpu(), F.conv2d(torch.cat([x, hn], 1), w).detach(), 1e-4, "out with fused BatchNorm")
    dh = ops.new_view(N, H, H, 32)
    dw = torch.zeros(33, device=DEV)
    ops.out_conv_bwd(x.to(DEV), hv, w.detach().reshape(33).to(DEV), dout.to(DEV), dh, dw)
    close(from_view(dh), hr.grad, what="dh")
    close(dw.cpu(), w.grad.reshape(33), 1e-4, "dw")


def test_spectral_norm_sigma_and_backward():
    R, K = 128, 576
    Wm = torch.randn(R, K, generator=torch.Generator().manual_seed(7)) * 0.05
    u = F.normalize(torch.randn(R, generator=torch.Generator().manual_seed(8)), dim=0)
    v = F.normalize(torch.randn(K, generator=torch.Generator().manual_seed(9)), dim=0)
    v1 = F.normalize(Wm.t() @ u, dim=0, eps=1e-12)
    u1 = F.normalize(Wm @ v1, dim=0, eps=1e-12)
    Wr = Wm.clone().requires_grad_(True)
    sigma = torch.dot(u1, Wr @ v1)
    Wsn = Wr / sigma
    dWsn = torch.randn(R, K, generator=torch.Generator().manual_seed(10))
    Wsn.backward(dWsn)
    ud, vd = u.to(DEV), v.to(DEV)
    sg = torch.zeros(1, device=DEV)
    scratch = torch.full((K + R,), float("nan"), device=DEV)
    ops.sn_sigma(Wm.to(DEV), R, K, ud, vd, sg, scratch, True)
    close(ud.cpu(), u1, 1e-5, "u")
    close(vd.cpu(), v1, 1e-5, "v")
    close(sg.cpu(), sigma.detach().reshape(1), 1e-5, "sigma")
    sg2 = torch.zeros(1, device=DEV)
    ops.sn_sigma(Wm.to(DEV), R, K, ud, vd, sg2, scratch, False)
    close(sg2.cpu(), sigma.detach().reshape(1), 1e-5, "sigma (no power iteration)")
    dW = torch.zeros(R, K, device=DEV)
    ops.sn_backward(dWsn.to(DEV), Wm.to(DEV), ud, vd, sg, R, K, dW, scratch[:1])
    close(dW.cpu(), Wr.grad, 1e-4, "dW")


def test_spectral_norm_job_tables_match_single_layer_calls():
    """All layers in one launch per phase = the per-layer entry points (same arithmetic; summation order of the
    sigma-gradient dot differs: 1e-5)."""
    g = torch.Generator().manual_seed(9)
    jobs = ops.SnJobs()
    ref = []
    for R, K in [(64, 9), (128, 576), (512, 4608), (40, 72)]:
        W = torch.randn(R, K, generator=g).to(DEV) * K ** -0.5
        u = torch.nn.functional.normalize(torch.randn(R, generator=g), dim=0).to(DEV)
        v = torch.nn.functional.normalize(torch.randn(K, generator=g), dim=0).to(DEV)
        dWsn = torch.randn(R, K, generator=g).to(DEV)
        u1, v1, s1 = u.clone(), v.clone(), torch.zeros(1, device=DEV)
        scratch = torch.zeros(K + R + 2, device=DEV)
        ops.sn_sigma(W, R, K, u1, v1, s1, scratch, True)
        dW1 = torch.zeros(R, K, device=DEV)
        ops.sn_backward(dWsn, W, u1, v1, s1, R, K, dW1, scratch[:1])
        s0 = torch.zeros(1, device=DEV)
        ops.sn_sigma(W, R, K, u1.clone(), v1.clone(), s0, scratch, False)     # sigma with the stored u, v
        u2, v2, s2, dW2 = u.clone(), v.clone(), torch.zeros(1, device=DEV), torch.full((R, K), 3.0, device=DEV)
        jobs.add(W, u2, v2, s2, R, K, dWsn, dW2)
        jobs.keep += [W, dWsn]                                   # the table holds raw addresses only
        ref.append((u1, v1, s1, dW1, s0, u2, v2, s2, dW2))
    jobs.finalize(DEV)
    ops.sn_sigma_jobs(jobs, power_iter=True)
    ops.sn_backward_jobs(jobs)
    torch.cuda.synchronize()
    for u1, v1, s1, dW1, s0, u2, v2, s2, dW2 in ref:
        close(u2.cpu(), u1.cpu(), 1e-6, "u"); close(v2.cpu(), v1.cpu(), 1e-6, "v"); close(s2.cpu(), s1.cpu(), 1e-6, "sigma")
        close(dW2.cpu(), dW1.cpu(), 1e-5, "dW")
    ops.sn_sigma_jobs(jobs, power_iter=False)
    torch.cuda.synchronize()
    for u1, v1, s1, dW1, s0, u2, v2, s2, dW2 in ref:
        close(s2.cpu(), s0.cpu(), 1e-6, "sigma without power iteration")


def test_mse_tap_and_loss_scalars():
    n = 4096 * 3
    a, b = r(n, seed=29), r(n, seed=30)
    loss = torch.zeros(1, device=DEV)
    g = torch.ones(n, dtype=torch.bfloat16, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    ops.mse_tap(ad.data_ptr(), bd.data_ptr(), n, 1.0 / n, loss, g.data_ptr(), 0.25, accumulate=True)
    close(loss.cpu(), ((a - b) ** 2).mean().reshape(1), 1e-4, "mse")
    close(g.float().cpu(), 1.0 + 0.25 * (a - b), what="grad")
    sc = torch.zeros(16, device=DEV)
    sc[0], sc[1], sc[5], sc[6] = 0.3, -2.0, 0.7, 0.4
    ops.dloss_scalars(sc, 1000)
    ops.gloss_scalars(sc, 0.3, 0.5, 0.2)
    s = sc.cpu()
    assert abs(s[2] - 0.7) < 1e-6 and abs(s[3] + 1e-3) < 1e-9 and s[4] == 0
    assert abs(s[7] - 0.3) < 1e-6 and abs(s[8] - 3.0) < 1e-6 and abs(s[9] - (0.21 + 0.15 + 0.6)) < 1e-6


def test_misc_elementwise():
    N, C, H = 2, 32, 4
    a, y = r(N, C, H, H, seed=31), r(N, C, H, H, seed=32)
    av, yv = to_view(a), to_view(y)
    o = ops.new_view(N, H, H, C, zero=True)
    ops.ew(ops.EW_ACTBWD, av, o, y=yv, slope=0.2)
    close(from_view(o), a * torch.where(y > 0, 1.0, 0.2), what="actbwd")
    ops.ew(ops.EW_ADD, av, o)
    close(from_view(o), a * torch.where(y > 0, 1.0, 0.2) + a, what="add")
    s = torch.tensor([0.7], device=DEV)
    ops.ew(ops.EW_AXPY, av, o, y=yv, s=s)
    close(from_view(o), 0.7 * a + y, what="axpy")
    cs = torch.zeros(C, device=DEV)
    ops.colsum(av, cs, torch.empty(ops.colsum_scratch_floats(C), device=DEV))
    close(cs.cpu(), a.sum((0, 2, 3)), 1e-3, "colsum")
    d = torch.zeros(1, device=DEV)
    ops.dot(av, yv, 0.5, d)
    close(d.cpu(), (0.5 * (a * y).sum()).reshape(1), 1e-3, "dot")
    ops.fill(o, 2.0, s=s)
    close(from_view(o), torch.full((N, C, H, H), 1.4), 5e-3, "fill")
    x32 = torch.randn(N, C, H, H, generator=torch.Generator().manual_seed(11))
    ops.nchw_to_nhwc(x32.to(DEV), o)
    close(from_view(o), x32, 5e-3, "nchw->nhwc")
    back = torch.empty(N, C, H, H, device=DEV)
    ops.nhwc_to_nchw(o, back)
    close(back.cpu(), from_view(o), 1e-7, "nhwc->nchw")


@pytest.mark.parametrize("N,C,H,W,c0", [(3, 640, 16, 16, 0), (2, 320, 8, 8, 64), (1, 8, 5, 7, 0), (5, 2048, 2, 3, 0),
                                        (2, 4096, 2, 2, 0), (64, 512, 4, 4, 0), (1, 24, 1, 1, 8)])
def test_column_sums(N, C, H, W, c0):
    """ried_colsum on whole tensors and on channel slices of wider ones (chunk columns that do not divide the block,
    fewer pixels than pixel lanes, more channels than one block's threads own)."""
    wide = r(N, C + c0 + 8, H, W, seed=71)
    v = to_view(wide).slice(c0, c0 + C)
    cs = torch.full((C,), 0.25, device=DEV)
    ops.colsum(v, cs, torch.full((ops.colsum_scratch_floats(C),), float("nan"), device=DEV))
    close(cs.cpu() - 0.25, wide[:, c0:c0 + C].sum((0, 2, 3)), 1e-3, "colsum")


def _ssim_ref(a, b, c1, c2):
    from oracle import ried_oracle as O
    return O.ssim_map(a, b, c1, c2)


@pytest.mark.parametrize("H,W", [(64, 64), (40, 72), (512, 512)])
def test_ssim_forward_backward(H, W):
    N = 2
    g = torch.Generator().manual_seed(12)
    ref = (torch.randn(N, 1, H, W, generator=g) * 520 - 470).clamp(-1024, 3072)
    a = (ref + torch.randn(N, 1, H, W, generator=g) * 30).clamp(-1024, 3072)
    ar = a.clone().requires_grad_(True)
    m = _ssim_ref(ref, ar, 1e-4, 9e-4)
    m.sum().backward()
    sums = torch.zeros(N, 4, device=DEV)
    gmap = torch.empty(3, N, H, W, device=DEV)
    ops.ssim_fwd(a.to(DEV), None, ref.to(DEV), 1e-4, 9e-4, sums, gmap=gmap)
    close(sums[:, 0].cpu(), m.detach().sum((1, 2, 3)), 2e-4, "ssim sum")
    close(sums[:, 1].cpu(), ((a - ref) ** 2).sum((1, 2, 3)), 1e-5, "sq err")
    da = torch.zeros(N, 1, H, W, device=DEV)
    ops.ssim_bwd(a.to(DEV), ref.to(DEV), gmap, 1.0, da)
    close(da.cpu(), ar.grad, 2e-3, "d ssim / d a")


def test_metrics_three_image_form():
    from oracle import ried_oracle as O
    N, H = 2, 128
    g = torch.Generator().manual_seed(13)
    y = torch.randn(N, 1, H, H, generator=g) * 520 - 470
    x = y + torch.randn(N, 1, H, H, generator=g) * 30
    p = y + torch.randn(N, 1, H, H, generator=g) * 10
    sums = torch.zeros(N, 4, device=DEV)
    ops.ssim_fwd(x.to(DEV), p.to(DEV), y.to(DEV), 16.0, 144.0, sums, clamp=(-160.0, 240.0))
    s = sums.cpu() / (H * H)
    for i in range(N):
        xi, yi, pi = [O.trunc(t[i, 0]) for t in (x, y, p)]
        o, q = O.compute_measure(xi, yi, pi, 400.0)
        assert abs(s[i, 0] - o[1]) < 2e-5 and abs(s[i, 2] - q[1]) < 2e-5
        assert abs(s[i, 1].sqrt() - o[2]) < 1e-3 * o[2] and abs(s[i, 3].sqrt() - q[2]) < 1e-3 * q[2]


@pytest.mark.parametrize("n,d,C", [(256, 32, 256), (64, 64, 512)])
def test_fused_attention_forward_backward(n, d, C):
    """ried_attn_fwd / ried_attn_bwd (Self_Attn, src/rinet.py:162-181, without its three projections) against torch
    autograd in fp32 on the bf16-rounded inputs.  Tolerances relative to each tensor's largest value: outputs 1.5e-2
    (bf16 storage of the attention weights and of O), gradients 3e-2, lse 1e-3 absolute."""
    N, side = 3, int(n ** 0.5)
    qkv = r(N, 2 * d + C, side, side, seed=71)
    qkv[:, :2 * d] *= 0.6                          # energies of a few units: a soft, not a one-hot, softmax
    qkv = qkv.to(torch.bfloat16).float()
    x = r(N, C, side, side, seed=72)
    dout = r(N, C, side, side, seed=73) * 0.1
    gamma = torch.tensor([0.7])
    assert ops.attn_fused_supported(n, d, C) and not ops.attn_fused_supported(n, d, C + 64)
    # torch reference on [N][n][channels]
    t = qkv.permute(0, 2, 3, 1).reshape(N, n, -1).clone().requires_grad_(True)
    xr = x.permute(0, 2, 3, 1).reshape(N, n, C)
    gr = gamma.clone().requires_grad_(True)
    q, k, v = t[:, :, :d], t[:, :, d:2 * d], t[:, :, 2 * d:]
    energy = q @ k.transpose(1, 2)
    att = torch.softmax(energy, dim=-1)
    Oref = att @ v
    out_ref = gr * Oref + xr
    do = dout.permute(0, 2, 3, 1).reshape(N, n, C)
    (out_ref * do).sum().backward()
    # device
    qv, xv, dv = to_view(qkv), to_view(x), to_view(dout)
    outv, Ov = ops.new_view(N, side, side, C, DEV), ops.new_view(N, side, side, C, DEV)
    lse = torch.full((N * n,), float("nan"), device=DEV)
    gd = gamma.to(DEV)
    ops.attn_fwd(qv, xv, outv, Ov, lse, gd, d)
    close(Ov.torch().float().cpu().reshape(N, n, C), Oref.detach(), 1.5e-2, "O")
    close(outv.torch().float().cpu().reshape(N, n, C), out_ref.detach(), 1.5e-2, "out")
    assert (lse.cpu().reshape(N, n) - torch.logsumexp(energy.detach(), dim=-1)).abs().max().item() < 1e-3
    dq = ops.new_view(N, side, side, 2 * d + C, DEV)
    dq.base.fill_(float("nan"))
    dgam = torch.full((1,), 2.0, device=DEV)
    ops.attn_bwd(qv, dq, dv, Ov, lse.data_ptr(), gd, dgam, d)
    got = dq.torch().float().cpu().reshape(N, n, -1)
    assert not torch.isnan(got).any()
    close(got[:, :, 2 * d:], t.grad[:, :, 2 * d:], 3e-2, "dv")
    close(got[:, :, d:2 * d], t.grad[:, :, d:2 * d], 3e-2, "dk")
    close(got[:, :, :d], t.grad[:, :, :d], 3e-2, "dq")
    # dgamma = sum dout * O is a sum of terms of both signs over O rounded to bf16 (2^-9 relative, independent errors)
    mag = (do * Oref.detach()).abs().sum().item()
    assert abs(dgam.item() - 2.0 - gr.grad.item()) <= 2e-4 * mag + 1e-3, (dgam.item(), gr.grad.item(), mag)
    # a sub-range of the images, no gamma gradient (the generator-loss pass through the discriminator)
    dq.base.fill_(float("nan"))
    ops.attn_bwd(qv.images(1, 3), dq.images(1, 3), dv.images(1, 3), Ov.images(1, 3), lse.data_ptr() + 4 * n, gd, None, d)
    got2 = dq.torch().float().cpu().reshape(N, n, -1)
    assert torch.isnan(got2[0]).all() and torch.equal(got2[1:], got[1:])


XF_CASES = [  # cin, cout, k, stride, pad, H, N, transposed
    (64, 64, (1, 7), 1, (0, 3), 16, 4, False),      # padding along W: transformed zeros must stay zero
    (48, 96, (7, 1), 1, (3, 0), 8, 8, False),       # channels per tap not a multiple of the 64-wide stage
    (128, 128, (3, 3), 2, (1, 1), 16, 4, False),    # the grid blocks' strided last layers
    (256, 32, (1, 1), 1, (0, 0), 8, 4, False),      # 32-row tile
    (64, 256, (1, 7), 1, (0, 3), 64, 8, False),     # 256-pixel tile on eight waves
    (224, 192, (3, 3), 1, (1, 1), 4, 8, False),     # a 3x3 layer forced onto the general kernel
    (128, 64, (4, 4), 2, (1, 1), 8, 4, True),       # transposed: four phases
]


@pytest.mark.parametrize("cin,cout,k,s,p,H,N,tr", XF_CASES)
def test_conv2d_with_batchnorm_input_transform(cin, cout, k, s, p, H, N, tr):
    """ried_conv2d_xf: the operand is a raw convolution output; BatchNorm (given mean / invstd) + ReLU are applied
    inside the kernel.  Against torch on the explicitly normalised tensor (rounded to bf16, as the kernel rounds what
    it feeds the matrix cores), with statistics of the result in the folded form."""
    x = r(N, cin, H, H, seed=70) * 2 + 0.3
    mean, invstd = r(cin, seed=71) * 0.5, r(cin, seed=72).abs() + 0.5
    gamma, beta = r(cin, seed=73) + 1.0, r(cin, seed=74) * 0.3
    xn = F.relu((x - mean[None, :, None, None]) * (gamma * invstd)[None, :, None, None] + beta[None, :, None, None])
    xn = xn.to(torch.bfloat16).float()
    if tr:
        w = r(cin, cout, *k, seed=75, scale=(cin * k[0] * k[1]) ** -0.5)
        ref = F.conv_transpose2d(xn, w, None, s, p)
        wp = pack(w.permute(1, 0, 2, 3))
    else:
        w = r(cout, cin, *k, seed=75, scale=(cin * k[0] * k[1]) ** -0.5)
        ref = F.conv2d(xn, w, None, s, p)
        wp = pack(w)
    Ho, Wo = ref.shape[-2:]
    yv = ops.new_view(N, Ho, Wo, cout)
    save = torch.stack([mean, invstd]).contiguous().to(DEV)
    stats = torch.zeros(ops.bn_fold_rows(), 2, cout, device=DEV)
    ops.conv2d_xf(to_view(x), wp, yv, k[0], k[1], s, s, p[0], p[1], save, gamma.to(DEV), beta.to(DEV), 1,
                  transposed=tr, stats=stats, stats_post=2)
    close(from_view(yv), ref, what="y")
    st = stats.sum(0)
    close(st[1].cpu(), (ref * ref).sum((0, 2, 3)), 5e-3, "sum of squares")
    # no activation: the affine map alone
    xa = ((x - mean[None, :, None, None]) * (gamma * invstd)[None, :, None, None] + beta[None, :, None, None])
    xa = xa.to(torch.bfloat16).float()
    ref2 = F.conv_transpose2d(xa, w, None, s, p) if tr else F.conv2d(xa, w, None, s, p)
    ops.conv2d_xf(to_view(x), wp, yv, k[0], k[1], s, s, p[0], p[1], save, gamma.to(DEV), beta.to(DEV), 0, transposed=tr)
    close(from_view(yv), ref2, what="y (no activation)")
