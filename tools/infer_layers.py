"""Every distinct convolution of the 512 x 512 inference batch (32 slices) - or, with the argument "step", of the
training step at batch 128 - under the library's own dispatch: kernel template, time alone (graph replay of 20
launches), TFLOP/s and algorithmic GB/s (input + output + weights once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops
STEP = sys.argv[1:] == ["step"]
sys.argv = sys.argv[:1]                  # (conv_tune reads a shape file named by its own first argument)
import conv_tune

shapes = conv_tune.gather_shapes() if STEP else conv_tune.gather_infer_shapes()
rows = []
for N, H, W, Cin, Cout, kh, kw, s, tr, cnt in shapes:
    ph, pw = ((kh - 1) // 2, (kw - 1) // 2) if kh != 4 else (1, 1)
    Ho, Wo = (H * s, W * s) if tr else ((H + 2 * ph - kh) // s + 1, (W + 2 * pw - kw) // s + 1)
    x = ops.new_view(N, H, W, Cin); x.base.normal_()
    y = ops.new_view(N, Ho, Wo, Cout)
    w = torch.randn(Cout * kh * kw * Cin, device="cuda").to(torch.bfloat16)
    rows_n = ops.conv2d_stat_rows(N, Ho, Wo, Cin, Cout, kh, kw, s, s, bool(tr))
    stats = torch.zeros(max(rows_n, ops.bn_fold_rows()) * 2 * Cout, device="cuda")
    f = lambda: ops.conv2d(x, w, y, kh, kw, s, s, ph, pw, transposed=bool(tr), stats=stats, stats_post=2)
    name = ops.conv2d_which(x, w, y, kh, kw, s, s, ph, pw, transposed=bool(tr), stats=stats, stats_post=2)
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * N * (H * W if tr else Ho * Wo) * Cin * Cout * kh * kw / (s * s if tr and kh == 4 else 1)
    by = 2.0 * (N * H * W * Cin + N * Ho * Wo * Cout + w.numel())
    rows.append((us * cnt, f"{N}x{H}x{W} {Cin:4d}->{Cout:4d} k{kh}x{kw} s{s} tr{tr} x{cnt}  {name:34s} {us:7.1f} us "
                           f"{fl / us / 1e6:6.0f} TF/s {by / us / 1e3:6.0f} GB/s"))
for t, r in sorted(rows, reverse=True):
    print(f"{t:8.1f}  {r}")
print("total %.1f us" % sum(t for t, _ in rows))
