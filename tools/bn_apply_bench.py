"""GB/s of bn_apply (and its accumulate form) on the activation shapes of the 512x512 inference pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops
for npix, C, ld in [(32 * 256 * 256, 32, 96), (32 * 128 * 128, 64, 192), (32 * 128 * 128, 32, 128), (32 * 128 * 128, 16, 16),
                    (32 * 128 * 128, 128, 128), (32 * 64 * 64, 64, 256), (32 * 64 * 64, 256, 256), (32 * 32 * 32, 192, 1024),
                    (32 * 256 * 256, 64, 96)]:
    N = 32
    hw = npix // N
    h = int(hw ** 0.5)
    x = ops.new_view(N, h, h, C); x.base.normal_()
    big = ops.new_view(N, h, h, ld, zero=True)
    y = big.slice(0, C) if ld != C else big
    save = torch.cat([torch.zeros(C), torch.ones(C)]).cuda()
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for acc in (False, True):
        f = lambda: ops.bn_apply(x, y, save, g, b, relu=True, accumulate=acc)
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                f()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        nbytes = npix * C * 2 * (3 if acc else 2)
        print(f"npix {npix:8d} C {C:4d} ld {ld:4d} acc {int(acc)}: {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s")
