"""Consumer convolution with the BatchNorm + ReLU of its producer applied on the way in (ried_conv2d_xf) against the
apply pass + plain convolution it replaces, on the in-chain layers of the 512 x 512 inference batch.
usage: python tools/xf_bench.py            (prints us per launch: apply, conv, apply + conv, xf conv; max |diff|)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops

# (N, H, W, Cin, Cout, kh, kw, stride, transposed)
SHAPES = [(32, 256, 256, 32, 64, 3, 3, 2, 0),       # conv2
          (32, 128, 128, 32, 48, 3, 3, 1, 0), (32, 128, 128, 48, 48, 3, 3, 1, 0), (32, 128, 128, 16, 32, 3, 3, 1, 0),
          (32, 128, 128, 48, 64, 3, 3, 2, 0), (32, 128, 128, 32, 64, 3, 3, 2, 0),
          (32, 64, 64, 64, 64, 7, 1, 1, 0), (32, 64, 64, 64, 64, 1, 7, 1, 0),
          (32, 64, 64, 128, 128, 1, 7, 1, 0), (32, 64, 64, 128, 128, 3, 3, 2, 0), (32, 64, 64, 64, 128, 3, 3, 2, 0),
          (32, 32, 32, 224, 192, 3, 3, 1, 0),
          (32, 32, 32, 128, 128, 1, 7, 1, 1), (32, 64, 64, 32, 48, 3, 3, 1, 1), (32, 64, 64, 48, 64, 4, 4, 2, 1)]


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tot = [0.0, 0.0]
for (N, H, W, Cin, Cout, kh, kw, s, tr) in SHAPES:
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    if kh == 4:
        ph = pw = 1
    Ho, Wo = (H * s, W * s) if tr else ((H + 2 * ph - kh) // s + 1, (W + 2 * pw - kw) // s + 1)
    raw = ops.new_view(N, H, W, Cin); raw.base.normal_()
    act = ops.new_view(N, H, W, Cin)
    y1, y2 = ops.new_view(N, Ho, Wo, Cout), ops.new_view(N, Ho, Wo, Cout)
    w = (torch.randn(Cout * kh * kw * Cin, device="cuda") * 0.1).to(torch.bfloat16)
    save = torch.cat([torch.randn(Cin, device="cuda") * 0.2, torch.rand(Cin, device="cuda") + 0.5])
    gamma, beta = torch.rand(Cin, device="cuda") + 0.5, torch.randn(Cin, device="cuda") * 0.3
    rows = ops.conv2d_stat_rows(N, Ho, Wo, Cin, Cout, kh, kw, s, s, bool(tr))
    stats = torch.zeros(rows * 2 * Cout, device="cuda")
    fa = lambda: ops.bn_apply(raw, act, save, gamma, beta, relu=1)
    fc = lambda: ops.conv2d(act, w, y1, kh, kw, s, s, ph, pw, transposed=bool(tr), stats=stats)
    fx = lambda: ops.conv2d_xf(raw, w, y2, kh, kw, s, s, ph, pw, save, gamma, beta, 1, transposed=bool(tr), stats=stats)
    ta, tc, tx = timed(fa), timed(fc), timed(fx)
    which = ops.conv2d_which(act, w, y1, kh, kw, s, s, ph, pw, transposed=bool(tr), stats=stats)
    d = (y1.torch().float() - y2.torch().float()).abs().max().item()
    ref = y1.torch().float().abs().max().item()
    tot[0] += ta + tc; tot[1] += tx
    print(f"{N}x{H}x{W} {Cin:4d}->{Cout:4d} {kh}x{kw} s{s} tr{tr}  apply {ta:6.1f}  conv {tc:6.1f}  sum {ta + tc:6.1f}  xf {tx:6.1f} us"
          f"   max|diff| {d:.3g} of {ref:.3g}   [{which}]", flush=True)
print(f"total: apply + conv {tot[0]:.1f} us, xf {tot[1]:.1f} us")
