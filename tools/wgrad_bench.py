"""Time the weight-gradient launches of the discriminator's 3x3 stride-2 layers at 256 images (graph replay of 20)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops
from reid_gan_amd.engine import _splitk      # (RIED_NO_WGRAD_ROWS=1: one tap per block)

CASES = [(256, 64, 64, 128, 2), (256, 32, 128, 256, 2), (256, 16, 256, 512, 2), (256, 8, 512, 512, 2),
         (128, 64, 64, 64, 1), (128, 32, 128, 128, 1), (128, 16, 48, 48, 1), (128, 32, 32, 64, 2)]
for N, H, Cg, Cp, s in CASES:
    Ho = H // s
    g = ops.new_view(N, H, H, Cg); g.base.normal_()
    p = ops.new_view(N, Ho, Ho, Cp); p.base.normal_()
    kp = N * Ho * Ho
    sk = int(os.environ.get("SPLITK", "0")) or _splitk(ops.conv2d_wgrad_blocks(Cg, Cp, 3, 3, kp), kp, whole_rounds=True)
    slab = torch.empty(sk * 9 * Cp * Cg, device="cuda")
    f = lambda: ops.conv2d_wgrad(g, p, 3, 3, s, s, 1, 1, slab, sk)
    f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    fl = 2.0 * kp * Cp * Cg * 9
    print(f"N{N} {H}x{H}x{Cg} -> {Ho}x{Ho}x{Cp} s{s} splitk {sk}: {us:.1f} us {fl / us / 1e6:.0f} TFLOP/s", flush=True)
