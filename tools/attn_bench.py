"""Time the fused self-attention forward and backward on the discriminator's two layers - 16x16 maps (n = 256, d = 32,
C = 256) and 8x8 maps (n = 64, d = 64, C = 512) - for 256 and 128 images, graph replay of 20 launches each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops


def timed(f, reps=20):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for nimg, d, C, H in ((256, 32, 256, 16), (128, 32, 256, 16), (256, 64, 512, 8), (128, 64, 512, 8)):
    qkv = ops.new_view(nimg, H, H, 2 * d + C); qkv.base.normal_(); qkv.base.mul_(0.3)
    x = ops.new_view(nimg, H, H, C); x.base.normal_()
    out, O = ops.new_view(nimg, H, H, C), ops.new_view(nimg, H, H, C)
    dout = ops.new_view(nimg, H, H, C); dout.base.normal_()
    dqkv = ops.new_view(nimg, H, H, 2 * d + C)
    lse = torch.empty(nimg * H * H, device="cuda")
    gamma, dgamma = torch.full((1,), 0.5, device="cuda"), torch.zeros(1, device="cuda")
    tf = timed(lambda: ops.attn_fwd(qkv, x, out, O, lse, gamma, d))
    tb = timed(lambda: ops.attn_bwd(qkv, dqkv, dout, O, lse.data_ptr(), gamma, dgamma, d))
    n = H * H
    fl_f = nimg * 2.0 * n * n * (d + C)
    fl_b = nimg * 2.0 * n * n * (2 * d + 2 * C + d + d)
    print(f"images {nimg} n {H * H}: forward {tf:.1f} us ({fl_f / tf / 1e6:.0f} TFLOP/s), backward {tb:.1f} us ({fl_b / tb / 1e6:.0f} TFLOP/s)",
          flush=True)
